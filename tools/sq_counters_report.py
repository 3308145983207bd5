"""profiles/r5_sq_counters.{txt,json} from the tables tools/r5_sq_counters.sh leaves in gpurun_out/<tag>/ (tools/pmc_summary.py format).
MFMA-pipe utilisation by COUNTER = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) (both summed over the launches of a kernel at
one grid size); effective shader clock = GRBM_GUI_ACTIVE per launch / the launch's duration is not derivable here (durations differ
between passes) — the GUI-active cycles per launch are listed so that it can be set against the bench's own timing of the same shape.
usage: python tools/sq_counters_report.py gpurun_out/r5sq profiles/r5_sq_counters"""
import collections, json, os, re, sys
src, dst = sys.argv[1], sys.argv[2]
LABEL = {"x3": "gemm_nt_f16x3_v3i_kernel (parity mode, 3 MFMAs per product), ViT-B/16 layer products at M = 252160: blocks 8865 = in_proj, 2955 = "
               "out_proj + c_proj, 11820 = c_fc",
         "attn": "attention_fwd_pair_kernel<8,64,SINGLE,1>, 1280 ViT-B/16 sequences x 12 heads (...Lb0E... = split-f16 operands, ...Lb1E... = single-pass f16)",
         "f16_in_proj": "gemm_nt_f16_pp_kernel (single-pass f16, persistent), in_proj [252160, 2304, 768] -> f16",
         "f16_out_proj": "gemm_nt_f16_pp_kernel, out_proj [252160, 768, 768] -> f16",
         "f16_c_fc": "gemm_nt_f16_pp_kernel<QuickGELU>, c_fc [252160, 3072, 768] -> f16",
         "f16_c_proj": "gemm_nt_f16_pp_kernel, c_proj [252160, 768, 3072] -> f16"}
rec = collections.defaultdict(lambda: collections.defaultdict(dict))
for fn in sorted(os.listdir(src)):
    m = re.match(r"sq_(.+)_(busy|clk)\.txt$", fn)
    if not m: continue
    for ln in open(os.path.join(src, fn)):
        q = re.search(r"^(\S+)\s+blocks=\s*(\d+) (\S+)\s+launches=\s*(\d+) sum=(\S+)", ln)
        if q: rec[m.group(1)][(q.group(1)[:60], int(q.group(2)))][q.group(3)] = (int(q.group(4)), float(q.group(5)))
out, lines = {}, ["# SQ counters of the round-5 build: rocprofv3 --kernel-trace --pmc (counters only; tools/r5_sq_counters.sh), one MI355X",
                  "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); MFMA instructions and GUI-active cycles are per launch", ""]
for grp, ks in rec.items():
    lines.append(f"## {LABEL.get(grp, grp)}")
    for (name, blocks), c in sorted(ks.items()):
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "SQ_BUSY_CU_CYCLES" not in c: continue
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (4.0 * c["SQ_BUSY_CU_CYCLES"][1])
        n_l = c["GRBM_GUI_ACTIVE"][0] / 8 if "GRBM_GUI_ACTIVE" in c else None           # (one record per XCD and launch)
        gui = c["GRBM_GUI_ACTIVE"][1] / c["GRBM_GUI_ACTIVE"][0] if "GRBM_GUI_ACTIVE" in c else None
        mf = c["SQ_INSTS_MFMA"][1] / (n_l or 1) if "SQ_INSTS_MFMA" in c and n_l else None
        lines.append(f"{name:60s} blocks={blocks:6d} mfma_busy={busy:.3f}  launches={int(n_l) if n_l else '?'}  MFMA instructions/launch={mf:.4g}  "
                     f"GUI-active cycles/launch={gui:.4g}" if gui else f"{name:60s} blocks={blocks:6d} mfma_busy={busy:.3f}")
        out.setdefault(grp, {})[f"{name}|blocks={blocks}"] = {"mfma_busy": busy, "gui_active_cycles_per_launch": gui, "mfma_instructions_per_launch": mf}
    lines.append("")
# aggregates the bench line reads
def agg(grp, pred=lambda k: True):
    num = den = 0.0
    for (name, blocks), c in rec.get(grp, {}).items():
        if pred((name, blocks)) and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c:
            num += c["SQ_VALU_MFMA_BUSY_CYCLES"][1]; den += 4.0 * c["SQ_BUSY_CU_CYCLES"][1]
    return num / den if den else None
summary = {"dominant_gemm_parity_mode_mfma_busy": agg("x3"),
           "attention_fwd_pair_split_f16_mfma_busy": agg("attn", lambda k: "Lb0E" in k[0]),
           "attention_fwd_pair_single_f16_mfma_busy": agg("attn", lambda k: "Lb1E" in k[0]),
           "f16_gemm_mfma_busy": {g: agg(g) for g in rec if g.startswith("f16_")}}
lines.append("## aggregates (cycles summed over the launches listed above)")
lines.append(json.dumps(summary, indent=1))
open(dst + ".txt", "w").write("\n".join(lines) + "\n")
json.dump({"source": "profiles/r5_sq_counters.txt (tools/r5_sq_counters.sh + tools/sq_counters_report.py)", "summary": summary, "kernels": out},
          open(dst + ".json", "w"), indent=1)
print("\n".join(lines))
