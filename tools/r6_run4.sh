timeout 2400 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config_a or config_b or config_c or config_d" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_groups.py -x -q -m gpu -k "workload_e or control_flow" 2>&1 | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6f_bench_n1.json 2> gpurun_out/r6f_bench_n1.err; tail -c 600 gpurun_out/r6f_bench_n1.err; python -c "
import json; d=json.load(open('gpurun_out/r6f_bench_n1.json')); r=d['roofline']
print({k:d[k] for k in ('value','value_device_resident','value_pcie_inclusive','ms_per_step')}); print({k:r[k] for k in ('achieved','frac','frac_algorithmic','hbm_frac_by_counters','traffic','traffic_stale','avg_launch_ms')}); print(d['pcie_inclusive']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
timeout 900 python bench.py --workload E --steps 5 --warmup 2 > gpurun_out/r6f_bench_E_n1.json 2> gpurun_out/r6f_bench_E_n1.err; tail -c 600 gpurun_out/r6f_bench_E_n1.err; python -c "
import json; d=json.load(open('gpurun_out/r6f_bench_E_n1.json')); print({k:d[k] for k in ('metric','value','ms_per_step','scaling')}, d['stage_ms'], d['cpu_baseline'], d['roofline']['frac'])"
