"""In-tree build of libpsp_hip.so (hipcc, gfx950 only).

    python path-space-pde-solver_amd/build.py [--force] [-j N]

One translation unit per (d, H) line of csrc/instances.def plus csrc/psp_api.hip, linked
into csrc/libpsp_hip.so next to the sources so the library travels with the tree.
hipcc cross-compiles without a GPU.
"""
import argparse
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libpsp_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def instances(fname="instances.def"):
    out = []
    with open(os.path.join(CSRC, fname)) as fh:
        for line in fh:
            m = re.match(r"\s*X\(\s*(\d+)\s*,\s*(\d+)\s*\)", line)
            if m:
                out.append((int(m.group(1)), int(m.group(2))))
    return out


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(src, obj, defs, deps, force):
    stamp = obj + ".sha"
    dig = _digest(deps, " ".join(FLAGS + defs))
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC] + FLAGS + defs + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))
    with open(stamp, "w") as fh:
        fh.write(dig)
    return obj, True


def build(force=False, jobs=None, verbose=True, extra_flags=(), lib_path=None, obj_dir=None, only=None):
    """extra_flags / lib_path / obj_dir / only=[(d,H)..] are for diagnostic variants (e.g.
    -DPSP_STAMPS into csrc/libpsp_hip_stamps.so); the default call builds the shipped library."""
    global FLAGS, OBJ, LIB
    saved = (FLAGS, OBJ, LIB)
    FLAGS = FLAGS + list(extra_flags)
    OBJ = obj_dir or OBJ
    LIB = lib_path or LIB
    try:
        return _build(force, jobs, verbose, only)
    finally:
        FLAGS, OBJ, LIB = saved


def _build(force, jobs, verbose, only):
    os.makedirs(OBJ, exist_ok=True)
    hdr = os.path.join(CSRC, "hjb_kernels.h")
    inc = os.path.join(HERE, "..", "include", "psp.h")
    idef = os.path.join(CSRC, "instances.def")
    inst_src = os.path.join(CSRC, "hjb_instance.hip")
    api_src = os.path.join(CSRC, "psp_api.hip")
    ghdr = os.path.join(CSRC, "gen_kernels.h")
    gdef = os.path.join(CSRC, "gen_instances.def")
    ginst_src = os.path.join(CSRC, "gen_instance.hip")
    whdr = os.path.join(CSRC, "hjbw_kernels.h")
    wdef = os.path.join(CSRC, "wide_instances.def")
    winst_src = os.path.join(CSRC, "hjbw_instance.hip")
    dhdr = os.path.join(CSRC, "hjbd_kernels.h")
    ddef = os.path.join(CSRC, "dense_instances.def")
    dinst_src = os.path.join(CSRC, "hjbd_instance.hip")
    # The split of an fp32 operand into its f16 pair (hjb_kernels.h split8) costs two instructions per value only when the SLP
    # vectoriser leaves its two fmas alone: every translation unit but the wide family's is compiled with it off (measured, same
    # box: d = 100 forward -1.8 %, diffusion iteration -3 %, others unchanged; wide d = 500 backward +30 % from spills -- it keeps
    # the round-3 form).
    NOSLP, CLASSIC = ["-fno-slp-vectorize"], ["-DPSP_SPLIT_CLASSIC=1"]
    tasks = [(api_src, os.path.join(OBJ, "psp_api.o"), NOSLP, [api_src, hdr, ghdr, whdr, dhdr, os.path.join(CSRC, "genl_kernels.h"), inc, idef, gdef, wdef, ddef])]
    for d, H in instances("dense_instances.def"):
        tasks.append((dinst_src, os.path.join(OBJ, "dnet_inst_%d_%d.o" % (d, H)),
                      ["-DPSP_D=%d" % d, "-DPSP_H=%d" % H] + NOSLP, [dinst_src, dhdr, whdr, hdr]))
    for d, H in instances():
        tasks.append((inst_src, os.path.join(OBJ, "inst_%d_%d.o" % (d, H)),
                      ["-DPSP_D=%d" % d, "-DPSP_H=%d" % H] + NOSLP, [inst_src, hdr, os.path.join(CSRC, "hjbs_kernels.h"), os.path.join(CSRC, "hjba_kernels.h"),
                       os.path.join(CSRC, "hjbq_kernels.h"), os.path.join(CSRC, "hjbx_kernels.h")]))
    for d, H in instances("wide_instances.def"):
        tasks.append((winst_src, os.path.join(OBJ, "wide_inst_%d_%d.o" % (d, H)),
                      ["-DPSP_D=%d" % d, "-DPSP_H=%d" % H] + CLASSIC, [winst_src, whdr, hdr, os.path.join(CSRC, "hjbc_kernels.h")]))
    wxinst_src = os.path.join(CSRC, "hjbwx_instance.hip")
    for d, H in instances("wide_instances.def"):
        if d <= 256:        # the split-product role-specialised backward of these instances: its own unit, SLP vectoriser off
            tasks.append((wxinst_src, os.path.join(OBJ, "widex_inst_%d_%d.o" % (d, H)),
                          ["-DPSP_D=%d" % d, "-DPSP_H=%d" % H] + NOSLP, [wxinst_src, os.path.join(CSRC, "hjbwx_kernels.h"), whdr, hdr]))
    for d, H in instances("gen_instances.def"):
        tasks.append((ginst_src, os.path.join(OBJ, "gen_inst_%d_%d.o" % (d, H)),
                      ["-DPSP_D=%d" % d, "-DPSP_H=%d" % H] + NOSLP, [ginst_src, ghdr, hdr]))
    jobs = jobs or min(6, os.cpu_count() or 2)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda t: _compile(t[0], t[1], t[2], t[3], force), tasks))
    objs = [r[0] for r in results]
    rebuilt = any(r[1] for r in results)
    # drop objects of instances that were removed from instances.def
    keep = set(os.path.basename(o) for o in objs)
    for f in os.listdir(OBJ):
        if f.endswith(".o") and f not in keep:
            os.remove(os.path.join(OBJ, f))
            rebuilt = True
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=None)
    a = ap.parse_args()
    build(force=a.force, jobs=a.j)
    sys.exit(0)
