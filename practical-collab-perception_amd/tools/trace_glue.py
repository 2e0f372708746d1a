"""Lists every torch (aten) op that launches device work during one inference step of a bench.py config, with the innermost line of
this repository on its Python stack -- the host-side glue around the HIP kernels (finalize, allocations with fills, copies).
usage: trace_glue.py [config=car]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'car'
    conf = bench.CONFIGS[name]
    cfg = bench.load_cfg(conf['yaml'])
    batch = int(cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU)
    model, _state, _ds = bench.build_model(cfg)
    dev = torch.device('cuda:0')
    model = model.to(dev).eval()
    for m in model.modules():
        if hasattr(m, 'materialize_pillars'):
            m.materialize_pillars = False
            m.reuse_buffers = True
    pts_np, metas = bench.make_points(conf, batch, 0)
    pristine = torch.from_numpy(pts_np).to(dev)

    def step():
        with torch.no_grad():
            return model({'points': pristine.clone(), 'batch_size': batch, 'metadata': metas})
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()
    tally = collections.Counter()
    dtime = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith('aten::') or ev.device_time_total <= 0 or ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
            continue
        where = '?'
        for fr in ev.stack:
            if (fr.startswith('pcdet/') or fr.startswith('pcp_amd/') or 'bench.py' in fr) and '_zeros_views' not in fr:
                where = fr
                break
        tally[(ev.name, where)] += 1
        dtime[(ev.name, where)] += ev.device_time_total
    for (op, where), n in sorted(tally.items(), key=lambda kv: -dtime[kv[0]]):
        print('%3d  %8.1f us  %-28s %s' % (n, dtime[(op, where)], op, where))
    print('total device-launching aten ops per step: %d, %.1f us of device time' % (sum(tally.values()), sum(dtime.values())))


if __name__ == '__main__':
    main()
