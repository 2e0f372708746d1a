""""Overlap buys 9 % at the price of 1.6 - 2.4 x longer kernels: streams contend for the same CUs; no CU-mask /
partition experiment is recorded."  This is that experiment: the headline's pipelined runner (two model replicas, three BEV-maker streams
each) with the replicas' streams created by hipExtStreamCreateWithCUMask, so that each replica owns a fixed part of the chip.

Layouts (256 CUs = 8 XCDs x 32; ROCm orders the mask bits XCD-interleaved on multi-XCD parts: bit i -> XCD i % 8, CU i / 8):
  none        the default streams (what bench.py runs)
  xcd_halves  replica 0 on XCDs 0-3, replica 1 on XCDs 4-7 (each replica keeps whole L2s to itself)
  cu_halves   replica 0 on the lower 16 CUs of every XCD, replica 1 on the upper 16 (both replicas use every L2)
  makers_half / makers_q   trunks share the whole chip, the BEV-maker streams are confined to half / a quarter of every XCD
Usage: python cu_mask_experiment.py [steps]"""
import ctypes
import json
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..', '..'))
for p in (REPO, os.path.join(REPO, 'practical-collab-perception_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from pcdet.models.pipelined import PipelinedDetector  # noqa: E402

hip = ctypes.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def layout(name, replica):
    allc = list(range(256))
    if name == 'xcd_halves':
        t = [i for i in allc if (i % 8) // 4 == replica]
        return t, t
    if name == 'cu_halves':
        t = [i for i in allc if (i // 8) // 16 == replica]
        return t, t
    if name == 'makers_q':
        return None, [i for i in allc if (i // 8) < 8]
    if name == 'makers_half':
        return None, [i for i in allc if (i // 8) < 16]
    return None, None


def run(name, steps):
    conf = bench.CONFIGS['disco']
    cfg = bench.load_cfg(conf['yaml'])
    model, _state, _ds = bench.build_model(cfg)
    model = model.cuda().eval()
    bench.set_pipeline_mode(model)
    pts, metas = bench.make_points(conf, 4, 0)
    pristine = torch.from_numpy(pts).cuda()
    bufs = [torch.empty_like(pristine), torch.empty_like(pristine)]
    pipe = PipelinedDetector(model, replicas=2)
    bufs[0].copy_(pristine)
    pipe.prepare(bufs[0], 4, metas)                   # creates the default streams
    if name != 'none':
        mains = []
        for r, m in enumerate(pipe.models):
            trunk, makers = layout(name, r)
            mains.append(masked_stream(trunk) if trunk is not None else pipe.mains[r])
            if makers is not None:
                m._maker_streams = [masked_stream(makers) for _ in range(3)]
        pipe.mains = mains
    for i in range(6):
        pipe.submit(bufs[i & 1], 4, metas, copy_from=pristine)
    pipe.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipe.submit(bufs[i & 1], 4, metas, copy_from=pristine)
    preds = pipe.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(layout=name, ms_per_step=round(1e3 * dt / steps, 4), frames_per_s=round(4 * steps / dt, 2),
                boxes=int(sum(p['pred_boxes'].shape[0] for p in preds)))


if __name__ == '__main__':
    # one layout per PROCESS (masked queues of an earlier layout stay allocated and slow the later ones down: the first version of this
    # script measured `none` at 10.1 ms first and 12.2 ms after the three masked layouts in the same process)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    if len(sys.argv) > 2:
        print(json.dumps(run(sys.argv[2], steps)), flush=True)
    else:
        import subprocess
        for name in ('none', 'xcd_halves', 'cu_halves', 'makers_half', 'makers_q', 'none'):
            subprocess.call([sys.executable, os.path.abspath(__file__), str(steps), name])
