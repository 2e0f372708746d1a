cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
python bench.py --config lately6 --steps 20 --warmup 5 > $O/bench_lately6.json 2> $O/bench_lately6.err; tail -c 2500 $O/bench_lately6.json; tail -3 $O/bench_lately6.err
python bench.py --steps 10 --warmup 3 --plugin-default --no-cpu-baseline > $O/bench_disco_plugin_default.json 2>&1; python -c "
import json; d=json.loads(open('$O/bench_disco_plugin_default.json').read().strip().splitlines()[-1]); print('plugin-default disco', d['value'], d['ms_per_step'])"
python bench.py --config car --steps 10 --warmup 3 --plugin-default --no-cpu-baseline > $O/bench_car_plugin_default.json 2>&1; python -c "
import json; d=json.loads(open('$O/bench_car_plugin_default.json').read().strip().splitlines()[-1]); print('plugin-default car', d['value'], d['ms_per_step'])"
python bench.py --config car --steps 20 --warmup 5 --graph --no-cpu-baseline > $O/bench_car_graph.json 2>&1; python -c "
import json; d=json.loads(open('$O/bench_car_graph.json').read().strip().splitlines()[-1]); print('graph car', d['value'], d['ms_per_step'])"
python bench.py --dist ring --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_disco_ring.json 2>&1; python -c "
import json; d=json.loads(open('$O/bench_disco_ring.json').read().strip().splitlines()[-1]); print('ring disco', d['value'], d['ms_per_step'])"
