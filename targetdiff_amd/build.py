"""Build libtargetdiff_hip.so (gfx950) in-tree with hipcc.  No torch dependency in the library.

    python -m targetdiff_amd.build          # incremental
    python -m targetdiff_amd.build --force
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(HERE, 'build')
LIB = os.path.join(LIBDIR, 'libtargetdiff_hip.so')
SOURCES = ['entry.cpp', 'pack.cpp', 'plan.cpp', 'forward.cpp', 'session.cpp', 'graph.hip', 'node.hip', 'gate.hip', 'edge16.hip', 'misc.hip', 'likelihood.hip', 'egnn.hip']
ARCH = 'gfx950'
# NB: the kNN distance uses __fmul_rn/__fadd_rn explicitly (td_dist2), so the default fp contraction is safe.
# -fvisibility=hidden: only the extern "C" entry points of include/targetdiff_hip.h (visibility push(default)) are exported
FLAGS = ['-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', f'--offload-arch={ARCH}', '-x', 'hip', '-Wall', '-Wno-unused-function']
# per-file additions.  edge16.hip: no SLP vectorisation -- it turns the 64 P_i adds of a row (and other adjacent fp32 adds) into
# v_pk_add_f32, which costs more than two plain adds next to the co-resident waves' matrix instructions (x2h key pass -2 %, h2x -1.5 %
# per C2 step, A/B in one gpurun call; MI355X_MICROARCH.md lists packed fp32 as an anti-lever beside MFMAs)
FILE_FLAGS = {'edge16.hip': ['-fno-slp-vectorize'],
              # node.hip: the LDS-DMA asm statement names m0 as clobbered; clang reports m0 as a reserved register for every instantiation
              'node.hip': ['-Wno-inline-asm']}


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (need ROCm >= 7.0)')
    return exe


def source_tag() -> str:
    """First 12 hex digits of the SHA-256 over every source file of the library and the compiler flags: compiled into the library
    (td_build_tag) so that a measurement -- a bench line, a PMC profile -- names the sources of the binary it ran, and a rebuild of
    the same sources (hipcc's output is not bit-reproducible) keeps the name."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.cpp', '.h', '.map'))]
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'targetdiff_hip.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(repr((FLAGS, sorted(FILE_FLAGS.items()))).encode())
    return h.hexdigest()[:12]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'targetdiff_hip.h'))
    objs, procs = [], []
    tag = source_tag()
    tag_file = os.path.join(OBJDIR, 'source_tag.txt')
    tag_changed = not os.path.exists(tag_file) or open(tag_file).read().strip() != tag
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [sp, os.path.abspath(__file__)] + headers) or (src == 'entry.cpp' and tag_changed):
            cmd = [hipcc()] + FLAGS + FILE_FLAGS.get(src, []) + ([f'-DTD_BUILD_TAG="{tag}"'] if src == 'entry.cpp' else []) + ['-c', sp, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f'[build] {src} FAILED', file=sys.stderr)
            if not verbose:
                print(out, file=sys.stderr)
    if failed:
        raise RuntimeError('hipcc compilation failed')
    with open(tag_file, 'w') as f:
        f.write(tag + '\n')
    if force or procs or _stale(LIB, objs + [os.path.join(CSRC, 'exports.map')]):
        cmd = [hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}', f'-Wl,--version-script,{os.path.join(CSRC, "exports.map")}',
               '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
