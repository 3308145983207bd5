"""Turns the raw counter tables of tools/r6_capture.sh (gpurun_out/r6/final/{traffic_*,sq_*}.txt) into the evidence files bench.py reads:
profiles/r6_gemm_hbm_traffic.{txt,json} (fabric bytes per launch of the dominant GEMM) and profiles/r6_sq_counters.{txt,json} (MFMA pipe
busy).  usage: python tools/r6_counters_summary.py [capture dir]"""
import json, os, re, sys
D = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6/final"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "gemm_nt_f16x3_v3i_kernel<false,false,false,4>"


def table(path):
    rows = {}
    for line in open(os.path.join(D, path)):
        m = re.search(r"blocks=\s*(\d+)\s+(\w+)\s+launches=\s*(\d+)\s+sum=([\d.e+]+)\s+per_launch=([\d.e+]+)", line)
        if m:
            rows[(int(m.group(1)), m.group(2))] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    return rows


def times(path):
    out = {}
    for line in open(os.path.join(D, path)):
        m = re.match(r"M=\d+\s+(\S+)\s+N=.*?:\s+([\d.]+) us", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


W = 768
lines = ["# round 6: fabric traffic of the dominant kernel %s per launch on the round-6 final build." % KERNEL,
         "# tools/r6_capture.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/gemm_epi_bench.py M (WRITE_SIZE in its own pass) over the",
         "# four GEMMs of a ViT-B/16 layer WITH their epilogues; counters in KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM section)",
         "# -> fetch_corrected = 2 x raw.  L2 memory-side counters: Infinity-Cache hits are inside them.  Launch times from the same script without counters.",
         "# (this file: tools/r6_counters_summary.py over the capture's raw tables)", "#",
         "# M        product       fetch_corrected_MB  write_MB  total_MB   us per launch"]
per_pass, algo = {}, {}
for M, images in ((252160, 20), (403456, 32)):
    f, w, t = table(f"traffic_fetch_{M}.txt"), table(f"traffic_write_{M}.txt"), times(f"traffic_time_{M}.txt")
    mt = (M + 255) // 256
    tot = 0.0
    for name, ntile, reps, tm in (("out/c_proj", 3, 2, "(%.1f, %.1f)" % (t["out_proj+res"], t["c_proj+res"])), ("in_proj", 9, 1, "%.1f" % t["in_proj"]),
                                  ("c_fc", 12, 1, "%.1f" % t["c_fc+gelu->pair"])):
        fe = f[(mt * ntile, "FETCH_SIZE")][2] * 2 * 1024 / 1e6
        wr = w[(mt * ntile, "WRITE_SIZE")][2] * 1024 / 1e6
        lines.append(f"  M={M} {name:14s} {fe:10.1f} {wr:9.1f} {fe + wr:9.1f}   {tm}")
        tot += reps * (fe + wr)
    # algorithmic bytes of the four launches: operand pairs in (4 B per element), outputs as the epilogue writes them
    a = 0.0
    for N, K, out_b, res in ((3 * W, W, 4, 0), (W, W, 4, 4), (4 * W, W, 4, 0), (W, 4 * W, 4, 4)):
        a += M * K * 4 + N * K * 4 + M * N * (out_b + res)
    per_pass[str(images)], algo[str(images)] = tot / 4 * 1e6, a / 4
    lines.append(f"  M={M} mean over the four launches of a layer: {tot / 4:.1f} MB per launch = {tot / 4 * 1e6 / (a / 4):.2f} x the algorithmic bytes ({a / 4 / 1e6:.1f} MB)")
open(os.path.join(ROOT, "profiles/r6_gemm_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"kernel": KERNEL, "bytes_per_launch_by_images_per_pass": per_pass, "algorithmic_bytes_per_launch_by_images_per_pass": algo,
           "source": "profiles/r6_gemm_hbm_traffic.txt: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/gemm_epi_bench.py on the "
                     "round-6 final build (tools/r6_capture.sh), FETCH x2 gfx950 correction; mean over the four launches of a layer; fabric-side counters "
                     "(Infinity-Cache hits included)"}, open(os.path.join(ROOT, "profiles/r6_gemm_hbm_traffic.json"), "w"))
print("\n".join(lines[-8:]))


def busy(rows, blocks=None):
    b = sum(v[1] for (bl, c), v in rows.items() if c == "SQ_VALU_MFMA_BUSY_CYCLES" and (blocks is None or bl in blocks))
    cu = sum(v[1] for (bl, c), v in rows.items() if c == "SQ_BUSY_CU_CYCLES" and (blocks is None or bl in blocks))
    return b / (4 * cu)


x3, fi, fc = table("sq_x3_busy.txt"), table("sq_f16_in_proj_busy.txt"), table("sq_f16_c_fc_busy.txt")
mt = (252160 + 255) // 256
res = {"x3_all": busy(x3), "x3_out_c_proj": busy(x3, {mt * 3}), "x3_in_proj": busy(x3, {mt * 9}), "x3_c_fc": busy(x3, {mt * 12}), "f16_in_proj": busy(fi), "f16_c_fc": busy(fc)}
txt = ["# round 6: SQ counters of the round-6 final build (tools/r6_capture.sh): rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES",
       "# over tools/gemm_epi_bench.py 252160 (parity-mode dominant GEMM, the four products of a ViT-B/16 layer at 20 images per pass) and tools/gemm_f16_bench.py (single-pass f16 GEMM).",
       "# MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), cycles summed over the launches.  (tools/r6_counters_summary.py)",
       f"parity mode, gemm_nt_f16x3_v3i_kernel, all four products: {res['x3_all']:.4f}", f"   out_proj + c_proj: {res['x3_out_c_proj']:.4f}", f"   in_proj: {res['x3_in_proj']:.4f}",
       f"   c_fc: {res['x3_c_fc']:.4f}", f"single-pass f16, gemm_nt_f16_pp_kernel, in_proj: {res['f16_in_proj']:.4f}", f"single-pass f16, gemm_nt_f16_pp_kernel, c_fc: {res['f16_c_fc']:.4f}", "", "raw:"]
for p in ("sq_x3_busy.txt", "sq_f16_in_proj_busy.txt", "sq_f16_c_fc_busy.txt"):
    txt += [l.rstrip() for l in open(os.path.join(D, p)) if "blocks=" in l]
open(os.path.join(ROOT, "profiles/r6_sq_counters.txt"), "w").write("\n".join(txt) + "\n")
print("\n".join(txt[3:9]))
json.dump({"summary": {"dominant_gemm_parity_mode_mfma_busy": res["x3_all"],
                       "per_product": {"out_proj + c_proj": res["x3_out_c_proj"], "in_proj": res["x3_in_proj"], "c_fc": res["x3_c_fc"]},
                       "f16_gemm_mfma_busy": {"in_proj": res["f16_in_proj"], "c_fc": res["f16_c_fc"]}}}, open(os.path.join(ROOT, "profiles/r6_sq_counters.json"), "w"))
