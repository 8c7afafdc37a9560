"""Random-shape check of dcn_fwd_b2_kernel against the torch-CPU oracle (CN_DCN_B2_MIN_HW=0 puts every map on it): python tools/dev/b2_fuzz.py [n] [seed]"""
import os, random, sys
os.environ.setdefault("CN_DCN_B2_MIN_HW", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import dcn_bm_check as c
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for i in range(n):
    N, H, W = rnd.choice([1, 2, 3]), rnd.randint(3, 70), rnd.randint(3, 70)
    Ci, sigma = rnd.choice([64, 64, 128, 256]), rnd.choice([0.0, 0.3, 1.0, 2.5, 6.0])
    worst = max(worst, c.check(N, H, W, Ci, 64, sigma, seed=i))
print("worst rel err over", n, "random configurations:", worst)
assert worst < 8e-3
