"""Compiles dkt_stereo_amd/csrc/*.hip into dkt_stereo_amd/lib/libdktstereo.so for
gfx950 with hipcc (cross-compiles without a GPU).  In-tree on purpose: the
built library travels with the repository snapshot to the GPU box."""
import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libdktstereo.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
          "-ffp-contract=off",  # the sampler arithmetic is reproduced bit for bit
          # NO packed fp32 math anywhere: the SLP vectoriser turns neighbouring scalar fp32 operations into
          # v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, and conv3x3_few_kernel's packed accumulators came back wrong (low
          # halves, lanes 48..63) whenever its block shared a CU with a block of an MFMA kernel -- 1000 of 1200 launches,
          # 0 of 1200 with scalar math.  The mechanism is not reproducible in isolation, so the rule is library-wide and
          # check_no_packed_fp32() below enforces it on the linked code objects.
          "-fno-slp-vectorize", "-fno-vectorize",      # (the loop vectoriser packs pairs of iterations the same way)
          "-Wno-pass-failed"]
FLAGS = CFLAGS + ["-shared"]       # single-command form: hipcc FLAGS csrc/*.hip -o libdktstereo.so
OBJ_DIR = os.path.join(LIB_DIR, "obj")
# Per-source flags.
# conv_c8: its step and epilogue are fully unrolled by construction (accumulators and fragments must stay in registers: a loop
# the unroller gives up on indexes them dynamically and sends the 128 accumulators to scratch -- 540 us instead of 280);
# the epilogue's body exceeds the default threshold of `#pragma unroll`.
EXTRA_FLAGS = {"conv_c8": ["-mllvm", "-pragma-unroll-threshold=100000"],
               "gru_c8": ["-mllvm", "-pragma-unroll-threshold=100000"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(_HERE, "..", "include", "dktstereo.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one object per source, compiled in parallel; conv2d.hip (the long pole: dozens of kernel
    # instantiations) is compiled as four translation units selected by -DCONV_TU_PASSES
    jobs = []
    for src in sources():
        stem = os.path.splitext(os.path.basename(src))[0]
        if stem == "conv2d":
            for tu in (0, 1, 2, 3):
                jobs.append((src, os.path.join(OBJ_DIR, "conv2d_tu%d.o" % tu), ["-DCONV_TU_PASSES=%d" % tu]))
        else:
            jobs.append((src, os.path.join(OBJ_DIR, stem + ".o"), EXTRA_FLAGS.get(stem, [])))

    def compile_one(job):
        src, obj, extra = job
        cmd = [HIPCC] + CFLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        return job, subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(len(jobs), int(os.environ.get("DKT_BUILD_JOBS", os.cpu_count() or 4))))
    with ThreadPoolExecutor(workers) as pool:
        results = list(pool.map(compile_one, jobs))
    for (src, obj, extra), res in results:
        if res.returncode != 0:
            sys.stderr.write(res.stdout)
            raise RuntimeError("hipcc failed compiling %s %s" % (os.path.basename(src), " ".join(extra)))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for (_, obj, _), _ in results] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed linking libdktstereo.so")
    bad = check_no_packed_fp32([obj for (_, obj, _), _ in results])
    if bad:
        raise RuntimeError("packed fp32 math in device code (DESIGN 3.4 forbids it):\n  " +
                           "\n  ".join("%s: %d x %s" % (k, n, op) for k, op, n in bad[:20]))
    return LIB


PACKED_FP32 = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def check_no_packed_fp32(objects):
    """Disassembles the gfx950 code object of every compiled source and returns [(kernel, opcode, count)] for every
    packed fp32 arithmetic instruction found (empty = the rule of DESIGN 3.4 holds).  Skipped (returns []) when
    llvm-objdump is unavailable."""
    import collections
    import re
    import shutil
    import tempfile
    if not os.path.exists(OBJDUMP):
        return []
    found = collections.Counter()
    tmp = tempfile.mkdtemp(prefix="dkt_pk_")
    try:
        for obj in objects:
            local = os.path.join(tmp, os.path.basename(obj))
            shutil.copy(obj, local)
            subprocess.run([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
            for f in sorted(os.listdir(tmp)):
                if not f.startswith(os.path.basename(obj) + ".") or "gfx950" not in f:
                    continue
                dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
                kernel = "?"
                for line in dis.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                    if m:
                        kernel = m.group(1)
                        continue
                    for op in PACKED_FP32:
                        if op in line:
                            found[(kernel, op)] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return sorted((k, op, n) for (k, op), n in found.items())


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
