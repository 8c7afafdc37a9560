"""Development check of the gather-sample DCNv2 kernels (csrc/dcn_gs.hip): the entry points on random inputs at several offset scales
(zero, sub-pixel, >= 2 px = the spare-slot path, 6 px = several passes per tile) and odd image sizes against oracle/dcn_ref.py on the
CPU, then isolated timings at the bench shape (run again with CN_DISABLE_DCN_GS=1 for the blend-matrix kernels).
    python tools/attic/dcn_gs_check.py [fwd|dom|dw|all] [time]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dcn_bm_check as bm  # noqa: E402

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    print("CN_DISABLE_DCN_GS =", os.environ.get("CN_DISABLE_DCN_GS"))
    shapes = [(2, 16, 32), (1, 13, 21), (3, 24, 40), (1, 8, 8), (2, 5, 7), (1, 40, 9)]
    sigmas = (0.0, 0.5, 1.0, 1.5, 2.5, 6.0)
    if what in ("fwd", "all"):
        worst = 0.0
        for (N, H, W) in shapes:
            for sigma in sigmas:
                worst = max(worst, bm.check(N, H, W, 64, 64, sigma))
        print("worst rel err fwd", worst)
    if what in ("dom", "all"):
        worst = 0.0
        for (N, H, W) in shapes:
            for sigma in sigmas:
                worst = max(worst, bm.check_dom(N, H, W, 64, sigma))
        print("worst rel err dom", worst)
    if what in ("dw", "all"):
        worst = 0.0
        for (N, H, W) in shapes:
            for sigma in sigmas:
                worst = max(worst, bm.check_dw(N, H, W, sigma))
        print("worst rel err dW", worst)
    if "time" in sys.argv:
        import opbench
        os.environ["DCN_SHAPES"] = "1"
        opbench.bench_dcn()
