#!/usr/bin/env python3
import os, sys
os.environ.setdefault("SGP_CHAIN", "persistent")
os.environ["SGP_CHAIN_DUMP"] = "/tmp/chain_dump.bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gaussianprocessnode_amd import device as Dv
rng = np.random.default_rng(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = rng.normal(size=(n, n)); A = B @ B.T / n + np.eye(n)
npad = (n + 63) // 64 * 64
Ap = np.eye(npad); Ap[:n, :n] = A
Lr = np.linalg.cholesky(Ap)
nfail = 0
for rep in range(40):
    L = Dv.potrf(A)
    raw = open("/tmp/chain_dump.bin", "rb").read()
    np_, Tn = np.frombuffer(raw[:8], dtype=np.int32)
    Lg = np.frombuffer(raw[8:], dtype=np.float64).reshape(3, np_, np_).transpose(0, 2, 1)[0]
    d = np.abs(np.tril(Lg) - Lr)
    d[np.isnan(d)] = 9.0
    if d.max() > 1e-10:
        nfail += 1
        blocks = sorted({(int(i) // 16, int(j) // 16) for i, j in zip(*np.where(d > 1e-10))})
        print("rep", rep, "max", d.max(), "bad 16-blocks", blocks[:30], flush=True)
        if nfail <= 2:
            bi, bj = blocks[0]
            sub = d[16*bi:16*bi+16, 16*bj:16*bj+16]
            g = np.tril(Lg)[16*bi:16*bi+16, 16*bj:16*bj+16]; e = Lr[16*bi:16*bi+16, 16*bj:16*bj+16]
            for r in range(16):
                print("   " + "".join("x" if sub[r, c] > 1e-10 else "." for c in range(16)), "  got %.4f exp %.4f" % (g[r, 0], e[r, 0]))
print("n", n, "failures", nfail, "of 40")
