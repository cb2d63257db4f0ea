timeout -k 10 300 python - <<'P' 2>&1 | grep -v amdgpu.ids | tail -30
import sys, numpy as np
sys.argv = ["fuzz", "1", "1155", "v"]
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import fuzz_parity as f
import gpu_helpers as g
f.VERBOSE = True
orig = g.check_dp_state
def dbg(name, eng, orc, n, K=3, stream=0):
    q = K * 5
    a, b = eng.get_state("modes", (q, n), np.float32, stream=stream), orc.get_state("modes", (q, n), np.float32)
    na, nb = eng.get_state("nmodes", (n,), np.uint8, stream=stream), orc.get_state("nmodes", (n,), np.uint8)
    bad = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
    print("nmodes equal:", np.array_equal(na, nb), "bad entries:", len(bad))
    for (qq, i) in bad[:10]:
        print(" plane", qq, "(mode", qq // 5, "field", qq % 5, ") pixel", i, "gpu", a[qq, i], "oracle", b[qq, i], "nmodes", na[i], nb[i], "gpu mode row", a[(qq // 5) * 5:(qq // 5) * 5 + 5, i], "oracle", b[(qq // 5) * 5:(qq // 5) * 5 + 5, i])
    orig(name, eng, orc, n, K=K, stream=stream)
g.check_dp_state = dbg
rng = np.random.default_rng(1155)
print(f.one_case(rng, 1155))
P
