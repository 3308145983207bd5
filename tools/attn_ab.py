"""Interleaved in-process A/B of attention_pair.hip builds (rlcf_attention_debug) on the benchmark-shaped ViT-B/16 case:
python tools/attn_ab.py "0:var,0:var,..." [rounds] [n_seq]; var 1 = shipped, 0 = eager rescale, 8 = no MFMAs, 32 = no DMA,
40 = neither, 64 = no per-block arithmetic (streaming only); "old" = the round-2 kernel on f32 qkv.  Median / min per build (us)."""
import sys, os, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcf_amd import _lib as L
lib = L.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
variants = [("old", 0) if v == "old" else tuple(int(x) for x in v.split(":")) for v in (sys.argv[1] if len(sys.argv) > 1 else "0:1,0:0,old").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
n_seq = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
tok, W = (int(os.environ.get("AB_TOK", "197")), int(os.environ.get("AB_W", "768")))
prec = L.PREC_F16 if os.environ.get("AB_SINGLE") else L.PREC_F16X3
T = n_seq * tok
qkv = torch.randn(T, 3 * W, device=dev)
seqs = torch.tensor([[i * tok, tok, 0, 0] for i in range(n_seq)], dtype=torch.int32, device=dev)
pairs = torch.empty(T, 3 * W, device=dev)
L.check(lib.rlcf_split_pairs(qkv.data_ptr(), pairs.data_ptr(), T * 3 * W, prec, st()))
op = torch.empty(T, W, device=dev)
out32 = torch.empty(T, W, device=dev)
def run(): L.check(lib.rlcf_attention_fwd_pairs(pairs.data_ptr(), seqs.data_ptr(), n_seq, tok, W, None, op.data_ptr(), None, prec, st()))
def run_old(): L.check(lib.rlcf_attention_fwd(qkv.data_ptr(), seqs.data_ptr(), n_seq, tok, W, 0, out32.data_ptr(), None, prec, st()))
times = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        fn = run_old if v[0] == "old" else run
        if v[0] != "old": lib.rlcf_attention_debug(0, v[1])
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        if r > 0: times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
flops = 4.0 * tok * tok * 64 * (W // 64) * n_seq
for v in variants:
    med = statistics.median(times[v])
    print(f"build={str(v[0]) + ':' + str(v[1]):7s}: median {med:8.1f} us  min {min(times[v]):8.1f}  max {max(times[v]):8.1f}   {flops / med / 1e6:6.1f} TF", flush=True)
