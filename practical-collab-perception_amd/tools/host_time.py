import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'practical-collab-perception_amd'))
import torch, bench
from pcdet.models.pipelined import PipelinedDetector
conf = bench.CONFIGS['disco']; cfg = bench.load_cfg(conf['yaml']); batch = int(cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU)
model, _s, _d = bench.build_model(cfg); dev = torch.device('cuda:0'); model = model.to(dev).eval(); model.overlap_makers = True
for m in model.modules():
    if hasattr(m, 'materialize_pillars'):
        m.materialize_pillars = False; m.reuse_buffers = True; m.sparse_first_layer = True
pts_np, metas = bench.make_points(conf, batch, 0); pristine = torch.from_numpy(pts_np).to(dev)
bufs = [torch.empty_like(pristine), torch.empty_like(pristine)]
pipe = PipelinedDetector(model)
for i in range(5): pipe.submit(bufs[i & 1], batch, metas, copy_from=pristine)
pipe.flush(); torch.cuda.synchronize()
t0 = time.perf_counter(); per = []
for i in range(20):
    a = time.perf_counter(); pipe.submit(bufs[i & 1], batch, metas, copy_from=pristine); per.append(time.perf_counter() - a)
t_host = time.perf_counter() - t0
pipe.flush(); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print('host loop %.1f ms, all %.1f ms (%.2f ms/step); submit() wall per step: min %.2f median %.2f ms' % (t_host*1e3, t_all*1e3, t_all*1e3/20, min(per)*1e3, sorted(per)[10]*1e3))
# pure host cost: the same loop while the GPU is NOT the bottleneck is not observable directly; profile python time of one submit with cProfile
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): pipe.submit(bufs[i & 1], batch, metas, copy_from=pristine)
pr.disable(); pipe.flush()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
