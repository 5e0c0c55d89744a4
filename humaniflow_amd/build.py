"""Build libhumaniflow_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m humaniflow_amd.build [--force] [--experiments]

The .so is git-ignored but travels with the repo snapshot to the GPU box.  `--experiments` builds a SECOND library,
tools/libhumaniflow_amd_experiments.so, from the same sources with -DHMF_EXPERIMENTS: the product entry points plus the timing / ablation
variants declared in include/humaniflow_amd_experiments.h (most of which compute wrong results).  Only scripts under tools/ load it;
libhumaniflow_amd.so never contains them.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['so3_kernels.hip', 'smpl_kernels.hip', 'flow_pack.hip', 'flow_kernels.hip', 'flow_grad.hip', 'flow_generic.hip', 'proxy_kernels.hip', 'encoder_train.hip', 'conv_wgrad_wino.hip', 'conv_dgrad_s2.hip', 'conv_igemm.hip', 'conv_gemm.hip', 'conv_gemm_split.hip', 'conv_pair.hip', 'conv_patch.hip', 'conv_winograd.hip', 'stem_conv.hip', 'stem_wino.hip', 'train_loss.hip', 'eval_kernels.hip', 'heads_kernels.hip']
# compiled only into tools/libhumaniflow_amd_experiments.so (--experiments): measured negative results kept for their probes, not product code
EXPERIMENT_SOURCES = ['conv_winograd_nw.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + ['../../include/humaniflow_amd.h', '../../include/humaniflow_amd_experiments.h']
LIB = os.path.join(HERE, 'libhumaniflow_amd.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -packed-fp32-ops: no v_pk_{fma,mul,add}_f32.  Packed f32 VALU beside MFMAs is an anti-lever on gfx950 (MI355X guide, price
# list): the skinning epilogue of k_lbs_mfma was emitted as ~24 v_pk_fma_f32 per (row, vertex) and did not overlap the
# partner wave's MFMAs at all (tools/lbs_tune.py).  The x86 host pass of hipcc prints "'-packed-fp32-ops' is not a recognized feature for this
# target (ignoring feature)": harmless, the device pass honours it (5036 -> 0 v_pk_*_f32 in smpl_kernels).
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function',
         '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


STAMP = LIB + '.stamp'
LOCK = os.path.join(HERE, '.build.lock')
EXP_DIR = os.path.join(os.path.dirname(HERE), 'tools')
EXP_LIB = os.path.join(EXP_DIR, 'libhumaniflow_amd_experiments.so')


def source_hash(experiments=False):
    """Content hash of everything the library is built from (sources, headers, flags, this file): staleness is decided by content, not
    by mtimes -- the snapshot that carries the tree to the GPU box does not preserve them."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS + (['-DHMF_EXPERIMENTS'] if experiments else [])).encode())
    for d in sorted(SOURCES + (EXPERIMENT_SOURCES if experiments else [])) + sorted(HEADERS):
        with open(os.path.join(CSRC, d), 'rb') as f:
            h.update(d.encode() + b'\0' + f.read())
    return h.hexdigest()


def _stale(experiments=False):
    lib = EXP_LIB if experiments else LIB
    if not os.path.exists(lib) or not os.path.exists(lib + '.stamp'):
        return True
    try:
        with open(lib + '.stamp') as f:
            return f.read().strip() != source_hash(experiments)
    except OSError:
        return True


def build_library(force=False, verbose=False, experiments=False):
    """Compile every HIP source into one shared library; returns its path.  Cheap when nothing changed.  Safe under N ranks starting
    together: one process builds under an exclusive file lock, the others wait and find a fresh library; the link goes to a temporary
    file that is renamed into place, so a concurrent CDLL never sees a half-written .so."""
    lib = EXP_LIB if experiments else LIB
    if not force and not _stale(experiments):
        return lib
    import fcntl
    with open(LOCK, 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(experiments):             # another rank built it while we waited
                return lib
            return _build_locked(verbose, experiments)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose, experiments=False):
    want = source_hash(experiments)
    LIB = EXP_LIB if experiments else globals()['LIB']
    STAMP = LIB + '.stamp'
    flags = FLAGS + (['-DHMF_EXPERIMENTS'] if experiments else [])
    objdir = os.path.join(EXP_DIR, '.obj') if experiments else CSRC
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in SOURCES + (EXPERIMENT_SOURCES if experiments else []):
        o = os.path.join(objdir, s.replace('.hip', '.o'))
        objs.append(o)
        cmd = [HIPCC] + flags + ['-c', os.path.join(CSRC, s), '-o', o]
        if verbose:
            print(' '.join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (s, out.decode(errors='replace')))
        if verbose and out:
            print(out.decode(errors='replace'))
    tmp = '%s.%d.tmp' % (LIB, os.getpid())
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s' % r.stdout.decode(errors='replace'))
    os.replace(tmp, LIB)
    with open(STAMP + '.tmp', 'w') as f:
        f.write(want + '\n')
    os.replace(STAMP + '.tmp', STAMP)
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True, experiments='--experiments' in sys.argv))
