"""Summary of an attention_pair.hip trace (RLCF_ATTN_VAR=16 RLCF_ATTN_TRACE_FILE=f python tools/attn_bench.py): per-wave s_memtime
stamps -> mean cycles per phase (s_memtime ticks at 100 MHz on gfx950: reported in ticks and in us)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8, 32).astype(np.int64)
act = a[:, :7]                                  # 7 active waves of a 197-token sequence
ok = act[:, :, 20] > 0
print("workgroups", a.shape[0], "traced waves", int(ok.sum()))
def seg(i, j, name):
    d = (act[:, :, j] - act[:, :, i])[ok]
    print(f"{name:34s} mean {d.mean():9.1f}  p50 {np.median(d):9.1f}  p95 {np.percentile(d, 95):9.1f}  max {d.max():9d}")
seg(0, 1, "prologue (issue, Q, wait chunk 0)")
for c in range(4):
    b = 2 + 4 * c
    if c > 0: seg(b - 4 + (3 if c < 4 else 2), b, f"chunk {c}: vmcnt + barrier")
    else: seg(1, 2, "chunk 0: barrier")
    seg(b, b + 1, f"chunk {c}: DMA issue")
    seg(b + 1, b + 2, f"chunk {c}: sub 0")
    if c < 3: seg(b + 2, b + 3, f"chunk {c}: sub 1")
seg(14 + 2, 18, "(last sub -> loop end)")
seg(18, 19, "epilogue (normalise, split, issue)")
seg(19, 20, "stores drained")
seg(0, 20, "whole wave")
print("-- inside chunk 1 / block 1 --")
seg(24, 25, "K reads + QK MFMAs + row max")
seg(25, 26, "rescale, exp2, row sum")
seg(26, 27, "tt=0: V reads issued + P split")
seg(27, 28, "tt=0: wait V")
seg(28, 29, "tt=0 MFMAs issued, tt=1 V issue+split")
seg(29, 30, "tt=1: wait V")
seg(30, 31, "tt=1 MFMAs issued")
seg(24, 31, "block total")
w = a[:, :, 20].max(axis=1) - a[:, :, 0].min(axis=1)
print("workgroup span mean", w.mean(), "p95", np.percentile(w, 95))
t0 = a[:, :, 0][a[:, :, 0] > 0].min(); t1 = a[:, :, 20].max()
print("launch span (ticks)", t1 - t0)
