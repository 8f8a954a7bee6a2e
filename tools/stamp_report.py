import sys, numpy as np
d = np.loadtxt(sys.argv[1])
names = ["first half (steps 0..H-1)", "dma wait", "barrier", "dma issue", "second half + combine", "between iterations"]
if len(sys.argv) > 2 and sys.argv[2] == "alt":
    names = ["C: MFMAs", "barrier behind C", "O: emission", "O: DMA wait", "O: finishing+DMA+priming", "barrier behind O"]
tot = d[:, 2:].sum(1)
print("waves %d, mean stamped cycles per wave %.0f (s_memtime ticks)" % (len(d), tot.mean()))
for i, n in enumerate(names):
    print("  %-28s mean %8.0f  (%5.1f%%)   min %8.0f max %8.0f" % (n, d[:, 2 + i].mean(), 100 * d[:, 2 + i].mean() / tot.mean(), d[:, 2 + i].min(), d[:, 2 + i].max()))
for w in range(8):
    sel = d[d[:, 1] == w]
    print("  wave %d: " % w + " ".join("%7.0f" % sel[:, 2 + i].mean() for i in range(6)))
