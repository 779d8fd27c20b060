import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.float32).reshape(5, -1)
x = a[0]
ref = [np.log2(x.astype(np.float64)).astype(np.float32), np.exp2((x - np.float32(2)).astype(np.float64)).astype(np.float32),
       np.exp((-x).astype(np.float64)).astype(np.float32), (x / np.float32(440)).astype(np.float32)]
for name, got, r in zip(("log2", "exp2", "exp", "div440"), a[1:], ref):
    bad = got != r
    print(f"{name}: {bad.mean() * 100:.3f} % of {x.size} differ from the correctly rounded fp32 value", x[bad][:4], got[bad][:4], r[bad][:4])
