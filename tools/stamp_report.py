import sys, numpy as np
d = np.loadtxt(sys.argv[1])
names = ["first half (steps 0..H-1)", "dma wait", "barrier", "dma issue", "second half + combine", "between iterations"]
if len(sys.argv) > 2 and sys.argv[2] == "alt":
    names = ["C: MFMAs", "barrier behind C", "O: emission", "O: DMA wait", "O: finishing+DMA+priming", "barrier behind O"]
extra = d[:, 8:12] if d.shape[1] > 8 else None
pro = d[:, 12:] if d.shape[1] > 12 else None
d = d[:, :8]
tot = d[:, 2:].sum(1)
print("waves %d, mean stamped cycles per wave %.0f (s_memtime ticks)" % (len(d), tot.mean()))
for i, n in enumerate(names):
    print("  %-28s mean %8.0f  (%5.1f%%)   min %8.0f max %8.0f" % (n, d[:, 2 + i].mean(), 100 * d[:, 2 + i].mean() / tot.mean(), d[:, 2 + i].min(), d[:, 2 + i].max()))
for w in range(8):
    sel = d[d[:, 1] == w]
    print("  wave %d: " % w + " ".join("%7.0f" % sel[:, 2 + i].mean() for i in range(6)))
if extra is not None:
    e8 = extra[:, 2].astype(np.uint64)
    real = (e8 >> np.uint64(32)).astype(np.float64)
    extra[:, 2] = (e8 & np.uint64(0xffffffff)).astype(np.float64)
    if real.max() > 0:
        print("  whole kernel, s_memrealtime: mean %.2f us max %.2f us -> %.3f GHz" % (real.mean() / 100, real.max() / 100, extra[:, 3].mean() / (real.mean() * 10)))
    for i, n in enumerate(["prologue: ring init + bank registers", "prologue: first tiles staged + primed", "epilogue", "whole kernel"]):
        print("  %-40s mean %8.0f  min %8.0f max %8.0f" % (n, extra[:, i].mean(), extra[:, i].min(), extra[:, i].max()))
if pro is not None:
    print("  prologue stamps since the second prologue mark (mean over waves of group 0 | group 1):")
    for i, n in enumerate(["tiles issued", "bank loads issued (group 0)", "ring cleared", "vmcnt(0)", "barrier", "bank loads issued (group 1)"]):
        print("    %-30s %8.0f | %8.0f" % (n, pro[d[:, 1] < 4, i].mean(), pro[d[:, 1] >= 4, i].mean()))
