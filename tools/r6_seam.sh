#!/bin/bash
# round 6: image format 4 + pipelined groups through the file seam -- its tests, then the seam at cfg4 (warm / cold), then the self-cleaning arena A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_seam_sharded.py tests/test_gpu_configs.py -x -q -m gpu -k "seam or image or cli or cfg3_file" 2>&1 | tail -15 > gpurun_out/r6_seam_tests.txt; cat gpurun_out/r6_seam_tests.txt
timeout 1500 python tools/seam_bench.py cfg4 > gpurun_out/r6_seam_bench.txt 2>gpurun_out/r6_seam_bench.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/seam_bench_cfg4.json"))
for k in ("files_to_tables_cold_s","files_to_tables_warm_s","files_to_tables_warm_s_both","db_load_cold_s","db_load_warm_s","gaf_load_s","db_bin_gb","db_image_gb","cold_and_warm_tables_same_bytes","image_writing_run_s"): print(k, d.get(k))
print(d.get("phases_ms_warm"))
t=d["trace"]["wd_warm1"]
print("\n".join(l[:300] for l in t.split("\n") if "gaf_tokenize]   piece" not in l))
PY
tail -3 gpurun_out/r6_seam_bench.err | cut -c1-300
for sc in 1 0; do
PANTAX_COV_SELF_CLEAN=$sc timeout 900 python bench.py --no-cpu-baseline --no-gaf --no-hard --no-l1 --steps 10 > gpurun_out/r6b_cfg4_clean$sc.json 2>gpurun_out/r6b_cfg4_clean$sc.err
echo "== self clean $sc"; python3 tools/bench_summary.py gpurun_out/r6b_cfg4_clean$sc.json | grep -E "^value|^kernels|^roofline|^  " | cut -c1-900; tail -2 gpurun_out/r6b_cfg4_clean$sc.err | cut -c1-300
done
