"""Build libboxinst_hip.so in-tree with plain hipcc for gfx950 (no torch cpp_extension, no hipify)."""
from __future__ import annotations

import os
import shutil
import subprocess
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libboxinst_hip.so')
SOURCES = ['abi.hip', 'pairwise_op.hip', 'color_affinity.hip', 'mask_loss.hip', 'fused_eval.hip', 'dynamic_head.hip', 'dynamic_head_generic.hip', 'meanfield.hip', 'levelset.hip', 'tree_filter.hip', 'tree_filter_large.hip', 'sol_eval.hip']
HEADERS = ['common.hpp', 'image_device.hpp', 'loss_common.hpp', 'dynamic_head_device.hpp', 'srgb_lut.h', os.path.join('..', '..', 'include', 'boxinst_hip.h'),
           os.path.join('..', '..', 'include', 'boxinst_hip_dev.h')]
ARCH = 'gfx950'


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: libboxinst_hip.so can only be built with the ROCm toolchain')
    return exe


FLAGS = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         # -amdgpu-kernarg-preload-count: the first kernel arguments arrive in SGPRs with the wave instead of through a
         # scalar load from the kernarg segment (one dependent memory round trip less at the start of every wave)
         '-mllvm', '-amdgpu-kernarg-preload-count=16']


def command(extra: List[str] | None = None, out: str | None = None) -> List[str]:
    """The one-command form (every source in one hipcc call): what a maintainer would type; build() below compiles the
    same sources with the same flags, one object per source in parallel, and links them."""
    return [hipcc(), *FLAGS, '-shared', *(extra or []), '-o', out or LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile_and_link(out: str, extra: List[str], verbose: bool) -> None:
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory(prefix='bxi_build_') as tmp:
        def one(src: str) -> str:
            obj = os.path.join(tmp, src.replace('.hip', '.o'))
            cmd = [hipcc(), *FLAGS, *extra, '-c', os.path.join(CSRC, src), '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
            objs = list(pool.map(one, SOURCES))
        link = [hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', out, *objs]
        if verbose:
            print(' '.join(link))
        subprocess.run(link, check=True)


def build(force: bool = False, verbose: bool = False, extra: List[str] | None = None) -> str:
    """Compile every HIP source into boxinstseg_amd/lib/libboxinst_hip.so; returns its path."""
    if force or is_stale():
        import fcntl
        os.makedirs(LIB_DIR, exist_ok=True)
        with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as lk:      # one builder at a time (ranks of a multi-GPU run)
            fcntl.flock(lk, fcntl.LOCK_EX)
            if force or is_stale():
                # link into a temporary name and rename: a process that dlopens LIB_PATH meanwhile sees the old or the new
                # library, never a partially written one, and a failed compile leaves no truncated file behind
                tmp = os.path.join(LIB_DIR, f'.libboxinst_hip.{os.getpid()}.so.tmp')
                try:
                    _compile_and_link(tmp, list(extra or []), verbose)
                    os.replace(tmp, LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
