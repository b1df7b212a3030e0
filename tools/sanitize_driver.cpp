// tools/sanitize_driver.cpp — the host-side logic that runs WITHOUT a GPU (the group threads' rendezvous with its error and time-out paths,
// the dry run of the strip choreography and its checker) under ThreadSanitizer and AddressSanitizer + UBSan: tools/sanitize_host.sh.
// (GPU AddressSanitizer is not available on this pool; the device code is covered by the parity tests.)
#include <cstdio>
#include <cstring>
extern "C" {
int lbm_debug_group_pool(int n, int rounds, int fail_strip, int fail_round, int stall_strip, int stall_round, int stall_ms, long timeout_ms, int repeat, int* rendezvous_out);
int lbm_debug_choreography(int nx, int ny, const int* bounds2, int nstrips, int precision, int transport, const char* options, const int* calls2, int ncalls, int dump, char* out, int cap);
int lbm_debug_p2p_matching(int nx, int ny, const int* bounds2, int nranks, int precision, const char* options, const char* options_rank1, const int* calls2, int ncalls, char* out, int cap);
const char* lbm_last_error(void);
}
int main() {
    int passed = 0;
    int rc = lbm_debug_group_pool(8, 2000, -1, -1, -1, -1, 0, 0, 3, &passed);
    printf("clean: rc %d passed %d\n", rc, passed);
    rc = lbm_debug_group_pool(6, 40, 3, 17, -1, -1, 0, 0, 2, &passed);
    printf("fail: rc %d passed %d: %s\n", rc, passed, lbm_last_error());
    rc = lbm_debug_group_pool(4, 20, -1, -1, 2, 9, 700, 150, 1, &passed);
    printf("stall: rc %d passed %d: %s\n", rc, passed, lbm_last_error());
    int b[6] = {0, 100, 100, 100, 200, 100}, calls[2] = {40, 0};
    static char out[1 << 16];
    int bad = 0, runs = 0;
    const char* plans[] = {"fuse=3 pair_ty=12", "deep=1", "deep=3", "deep=7 arith=1", "deep=9"};
    for (const char* plan : plans)
        for (int dh = 0; dh < 3; ++dh)
            for (int ov = 0; ov < 3; ++ov)
                for (int transport = 0; transport < 2; ++transport) {
                    char opts[160];
                    snprintf(opts, sizeof(opts), "tune=0 nt=1 xcd=1 overlap=%d deep_halo=%d %s", ov, dh, plan);
                    rc = lbm_debug_choreography(256, 300, b, 3, 0, transport, opts, calls, 1, 0, out, sizeof(out));
                    bad += rc != 0; ++runs;
                }
    printf("choreography: %d dry runs, %d flagged or failed\n", runs, bad);
    int b8[16], bad2 = 0, runs2 = 0;
    for (int k = 0; k < 8; ++k) { b8[2 * k] = 128 * k; b8[2 * k + 1] = 128; }
    for (const char* plan : plans)
        for (int trim = 0; trim < 2; ++trim) {
            char opts[160];
            snprintf(opts, sizeof(opts), "tune=0 nt=1 xcd=1 overlap=1 deep_halo=1 halo_trim=%d trailing_pair=1 %s", trim, plan);
            rc = lbm_debug_p2p_matching(4096, 1024, b8, 8, 0, opts, nullptr, calls, 1, out, sizeof(out));
            bad2 += rc != 0; ++runs2;
        }
    printf("p2p matching: %d eight-rank dry runs, %d mismatching or failed\n", runs2, bad2);
    return 0;
}
