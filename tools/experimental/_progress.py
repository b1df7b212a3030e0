"""Progress record of the long GPU tools (round 5): every case writes `START <case>` before it touches the GPU and `ok` / `MISMATCH` after,
straight to a file under gpurun_out/ with a flush — never through a `| grep | tail` pipe. A run that is killed for being silent
(seven minutes, an empty log) then leaves the case it was in on record; LBM_TRACE (set here unless the caller did)
adds the library call it was in."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Progress:
    def __init__(self, tool, tag):
        d = os.path.join(ROOT, "gpurun_out", "progress")
        os.makedirs(d, exist_ok=True)
        self.path = os.path.join(d, f"{tool}_{tag}.log")
        os.environ.setdefault("LBM_TRACE", os.path.join(d, f"{tool}_{tag}.trace"))      # (before the library is loaded)
        self.f = open(self.path, "a")
        self.t0 = time.time()
        self.say(f"# {tool} {' '.join(sys.argv[1:])} pid {os.getpid()}")

    def say(self, text):
        self.f.write(f"{time.time() - self.t0:9.3f} {text}\n")
        self.f.flush()
        os.fsync(self.f.fileno())

    def start(self, case):
        self.say(f"START {case}")

    def done(self, verdict="ok"):
        self.say(verdict)
