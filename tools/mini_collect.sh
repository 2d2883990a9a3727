TAG=${1:-round4a}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
line() { grep "^{" | tail -1; }
timeout 400 python tools/measure_traffic.py $O/${TAG}_traffic.json > $O/traffic.log 2>&1
cp $O/${TAG}_traffic.json profiles/round6_traffic.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | line > $O/${TAG}_bench_n1.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | line > $O/${TAG}_bench_under_rocprof.json
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
bash tools/pmc_kernels.sh 1 $TAG/pmc_f32 > $O/${TAG}_pmc_f32.txt 2>&1
rm -rf $O/trace $O/pmc_f32
cat $O/${TAG}_bench_kernel_stats.csv | head -12
cat $O/${TAG}_pmc_f32.txt | head -14
python -c "
import json; d=json.load(open('$O/${TAG}_bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['step_hbm'], d['cpu_baseline']['value'], d.get('dropin',{}).get('fused_device_f32'), d.get('dropin',{}).get('literal_device_f32'), d.get('dropin',{}).get('literal_device_i16'), d.get('dropin',{}).get('fused_host_f32'), d.get('dropin',{}).get('literal_host_f32'))"
