"""Phase times of mx_bank_kernel from a -DTSR_STAMP build (tools/build_stamp.sh; TSR_STAMP_OUT=file)."""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.float64)
d = d[d[:, 2] > 0]
t = d[:, 2:8]
x = d[:, 8:11]   # finer stamps: 6 loads landed, 7 own sums picked, 8 first tile finished
names = ["fill (bits -> nibbles in LDS, tile bases)", "barrier", "K loop", "partial sums -> LDS", "wait for the others' sums", "sum + epilogue"]
ph = np.diff(t, axis=1)
whole = t[:, 5] - t[:, 0]
real = d[:, 17] - d[:, 16]
print("waves %d; whole kernel per wave: mean %.0f max %.0f cycles; %.2f us (s_memrealtime) -> %.3f GHz" %
      (len(d), whole.mean(), whole.max(), real.mean() / 100, whole.mean() / (real.mean() * 10)))
for i, n in enumerate(names[:5]):
    print("  %-44s mean %7.0f  min %7.0f  max %7.0f" % (n if i != 3 else names[3] + " + barrier", ph[:, i].mean(), ph[:, i].min(), ph[:, i].max()))
for w in range(8):
    sel = ph[d[:, 1] == w]
    print("  wave %d: " % w + " ".join("%7.0f" % sel[:, i].mean() for i in range(5)))
if x.max() > 0:
    print("  inside the fill: loads landed after %.0f cycles (mean; max %.0f)" % ((x[:, 0] - t[:, 0]).mean(), (x[:, 0] - t[:, 0]).max()))
    print("  inside the finish: own sums picked %.0f, first tile finished after another %.0f, the rest %.0f" %
          ((x[:, 1] - t[:, 4]).mean(), (x[:, 2] - x[:, 1]).mean(), (t[:, 5] - x[:, 2]).mean()))
if d.shape[1] > 11 and d[:, 11].max() > 0:
    kc = d[:, 11]
    print("  K-blocks per wave (mean by wave id): " + " ".join("%.1f" % kc[d[:, 1] == w].mean() for w in range(8)) + "; cycles per K-block: " + " ".join("%.0f" % (ph[d[:, 1] == w, 2] / kc[d[:, 1] == w]).mean() for w in range(8)))
g0 = t[:, 0].min()
print("  first wave start -> last wave end: %.0f cycles; last start %.0f" % (t[:, 5].max() - g0, t[:, 0].max() - g0))
