"""Build oracle/_ref: the reference's OWN CPU rotated-IoU source compiled where it lies.

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


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
