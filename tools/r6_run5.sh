timeout 2400 python tests/fullsize/run_configs.py --configs A,B,C,D1 --out gpurun_out/r6g_configs_A_B_C_D1_full_queues.json > gpurun_out/r6g_configs.log 2>&1
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r6g_configs_A_B_C_D1_full_queues.json")):
    print(r["config"][:20], {k: r[k] for k in ("pois","oracle_sample","oracle_bit_exact","fftcc_oracle_same_integers","seq_flag_mismatches","seq_iteration_agreement","seq_max_abs_d_disp","seq_pois_over_1e4","seq_frac_within_1e4","seq_max_abs_d_zncc","oracle_seconds_fftcc_lanes_seq","seconds")}, "fma:", {k: r["fma"][k] for k in ("oracle_bit_exact","seq_max_abs_d_disp","seq_pois_over_1e4","seq_iteration_agreement")})
PY
