set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02/prof7; mkdir -p $O
timeout -k 10 900 tools/profile_round.sh $O 1 > $O/part1.log 2>&1
tail -3 $O/part1.log | cut -c1-100
