mkdir -p gpurun_out/mt3
RLCF_X3_MT3=0 timeout 600 python tools/gemm_mt3_check.py > gpurun_out/mt3/off.txt 2>&1; echo "off rc=$?"
RLCF_X3_MT3=1 timeout 600 python tools/gemm_mt3_check.py > gpurun_out/mt3/on.txt 2>&1; echo "on rc=$?"
grep ^SIG gpurun_out/mt3/off.txt > /tmp/a.txt; grep ^SIG gpurun_out/mt3/on.txt > /tmp/b.txt
wc -l /tmp/a.txt /tmp/b.txt
if diff /tmp/a.txt /tmp/b.txt > gpurun_out/mt3/sigdiff.txt; then echo "BIT-IDENTICAL"; else echo "SIG DIFF"; head -20 gpurun_out/mt3/sigdiff.txt; fi
paste <(grep "^MT3" gpurun_out/mt3/off.txt | awk '{print $2,$3,$4,$5,$6,$8}') <(grep "^MT3" gpurun_out/mt3/on.txt | awk '{print $(NF-4), $(NF-1)}' )
tail -3 gpurun_out/mt3/on.txt
