"""Build oracle/_ref: the reference's OWN CPU sources compiled where they lie (rotated IoU; round 5: points-in-boxes).

Recipe (runs only where /root/reference exists, i.e. in the CPU container; the GPU box uses the prebuilt .so
that travels with the snapshot):  g++ on /root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp + oracle/ref_bind.cpp
through torch.utils.cpp_extension (the source includes <torch/extension.h>).  The file also includes <cuda.h> and
<cuda_runtime_api.h>; this image ships genuine CUDA toolkit headers inside the triton wheel
(site-packages/triton/backends/nvidia/include), which are put on the include path -- no stand-in headers are written.
Outputs go only to oracle/_ref/ (git-ignored, NOT gpurun-ignored).
"""
import os
import sys

REF_SRC = '/root/reference/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')


def cuda_header_dir():
    import triton
    d = os.path.join(os.path.dirname(triton.__file__), 'backends', 'nvidia', 'include')
    if not os.path.isfile(os.path.join(d, 'cuda_runtime_api.h')):
        raise RuntimeError('no CUDA toolkit headers in this image: oracle/_ref is unbuildable here')
    return d


def build(verbose=False):
    build_roiaware(verbose)
    if not os.path.isfile(REF_SRC):
        return None
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, 'ref_iou3d_cpu.so')
    if os.path.isfile(so) and os.path.getmtime(so) >= os.path.getmtime(REF_SRC):
        return so
    from torch.utils.cpp_extension import load
    load(name='ref_iou3d_cpu', sources=[REF_SRC, os.path.join(HERE, 'ref_bind.cpp')],
         extra_include_paths=[cuda_header_dir()], extra_cflags=['-O2', '-w'],
         build_directory=OUT, verbose=verbose, with_cuda=False)
    return so


def load_ref():
    """Import the prebuilt module (works on the GPU box too, where /root/reference is absent)."""
    so = os.path.join(OUT, 'ref_iou3d_cpu.so')
    if not os.path.isfile(so):
        if build() is None:
            return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location('ref_iou3d_cpu', so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- round 5: roiaware_pool3d.cpp (points_in_boxes_cpu / check_pt_in_box3d_cpu, :123-166) -------------------------------------------------
# The file carries its own PYBIND11_MODULE and declares three CUDA launchers it never defines (they live in the .cu).  It is compiled ALONE,
# from where it lies, into a shared object whose three launcher symbols stay undefined: the module is imported with lazy binding
# (RTLD_LAZY), and only points_in_boxes_cpu -- which calls none of them -- is ever called.  No stand-in source is written.
ROI_SRC = '/root/reference/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp'


def build_roiaware(verbose=False):
    if not os.path.isfile(ROI_SRC):
        return None
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, 'ref_roiaware_pool3d.so')
    if os.path.isfile(so) and os.path.getmtime(so) >= os.path.getmtime(ROI_SRC):
        return so
    import subprocess
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = ['g++', '-O2', '-w', '-shared', '-fPIC', '-std=c++17', '-DTORCH_EXTENSION_NAME=ref_roiaware_pool3d', '-DTORCH_API_INCLUDE_EXTENSION_H',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ['-I' + d for d in ce.include_paths()] + ['-I' + sysconfig.get_paths()['include']]
    cmd += [ROI_SRC, '-o', so, '-L' + libdir, '-Wl,-rpath,' + libdir, '-ltorch', '-ltorch_cpu', '-lc10', '-ltorch_python']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return so


def load_roiaware():
    """the reference's roiaware_pool3d module (only points_in_boxes_cpu is usable); None where it cannot be built"""
    so = os.path.join(OUT, 'ref_roiaware_pool3d.so')
    if not os.path.isfile(so) and build_roiaware() is None:
        return None
    import importlib.util
    import torch  # noqa: F401
    flags = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY)                 # the CUDA launchers the file declares are never defined and never called
    try:
        spec = importlib.util.spec_from_file_location('ref_roiaware_pool3d', so)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(flags)
    return mod


def ref_points_in_boxes_cpu(points, boxes):
    """(M, 3) points, (T, 7) boxes float32 -> the reference's (T, M) int mask (margin 1e-2, roiaware_pool3d.cpp:123-166); None if unbuilt"""
    import numpy as np
    import torch
    mod = load_roiaware()
    if mod is None:
        return None
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32))
    p = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
    out = torch.zeros((b.shape[0], p.shape[0]), dtype=torch.int32)
    mod.points_in_boxes_cpu(b, p, out)
    return out.numpy()


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
    print(build_roiaware(verbose='-v' in sys.argv))
