"""diagnostic: where do the two store forms of attention_fwd_pair_kernel differ?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlcf_amd import _lib as L
lib = L.lib(); dev = torch.device("cuda:0"); st = lambda: torch.cuda.current_stream().cuda_stream
n_seq, tok, W = 3, 197, 128
T = n_seq * tok
torch.manual_seed(1)
qkv = torch.randn(T, 3 * W, device=dev) * 1.5
seqs = torch.tensor([[i * tok, tok, 0, 0] for i in range(n_seq)], dtype=torch.int32, device=dev)
for prec, name in ((L.PREC_F16X3, "pair"), (L.PREC_F16, "single")):
    pairs = torch.empty(T, 3 * W, device=dev)
    L.check(lib.rlcf_split_pairs(qkv.data_ptr(), pairs.data_ptr(), T * 3 * W, prec, st()))
    res = {}
    for v in ("0", "1"):
        os.environ["RLCF_ATTN_LINEST"] = v
        op = torch.full((T, W), 7.25, device=dev); lse = torch.full((T, W // 64), -3.0, device=dev)
        L.check(lib.rlcf_attention_fwd_pairs(pairs.data_ptr(), seqs.data_ptr(), n_seq, tok, W, None, op.data_ptr(), lse.data_ptr(), prec, st()))
        torch.cuda.synchronize()
        res[v] = (op.cpu().view(torch.int16).reshape(T, -1), lse.cpu())
    a, b = res["0"][0], res["1"][0]
    d = (a != b)
    print(name, "differing halves:", int(d.sum()), "of", d.numel(), "; lse equal:", torch.equal(res["0"][1], res["1"][1]))
    if d.any():
        rows = d.any(1).nonzero().flatten()
        print("  rows with differences:", rows[:40].tolist(), "... count", len(rows))
        r = int(rows[0]); cols = d[r].nonzero().flatten()
        print("  row", r, "cols", cols[:64].tolist())
        print("  pieces:", a[r, cols[:8]].view(torch.float16).tolist(), "lines:", b[r, cols[:8]].view(torch.float16).tolist())
        # is the lines row equal to some OTHER row of the pieces output?
        for rr in range(max(0, r - 40), min(T, r + 40)):
            if torch.equal(b[r], a[rr]): print("  lines row", r, "== pieces row", rr)
