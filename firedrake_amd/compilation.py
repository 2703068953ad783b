"""hipcc JIT + on-disk cache for wrapper kernels.

Mirror of pyop2/compilation.py:424-455 (``load``) and :527-611 (``make_so``): the
generated source is hashed together with the compiler identity and flags, compiled once
by a subprocess and cached on disk; later runs (and other ranks) just load the artefact.
Here the artefact is a gfx950 code object (``hipcc --genco``) loaded through
``fd_kernel_load`` (hipModuleLoad) instead of ``ctypes.CDLL``.
"""
import hashlib
import json
import os
import re
import subprocess
import tempfile

from .configuration import configuration

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


class CompilationError(RuntimeError):
    """pyop2/compilation.py:587-607 analogue."""


_version = None
stats = {"hipcc_runs": 0, "cache_hits": 0}      # wrappers compiled / found in the disk cache by this process


def compiler_version():
    """First line of ``hipcc --version`` (part of every cache key).  The answer is remembered next to the code objects under the
    identity of the compiler binary (path, size, mtime), so that a process whose wrappers are all cached starts no subprocess at
    all (30 ms of every first assemble); a different binary asks again."""
    global _version
    if _version is None:
        hipcc = configuration["hipcc"]
        ident, memo = None, os.path.join(configuration["cache_dir"], ".compiler_version.json")
        try:
            st = os.stat(os.path.realpath(hipcc))
            ident = f"{os.path.realpath(hipcc)}:{st.st_size}:{int(st.st_mtime)}"
            with open(memo) as fh:
                known = json.load(fh)
            if known.get("ident") == ident:
                _version = known["version"]
                return _version
        except (OSError, ValueError, KeyError):
            pass
        try:
            out = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
            _version = out.splitlines()[0] if out else "unknown"
        except OSError:
            _version = "missing"
        if ident and _version not in ("unknown", "missing"):
            try:
                os.makedirs(configuration["cache_dir"], exist_ok=True)
                fd, tmp = tempfile.mkstemp(dir=configuration["cache_dir"])
                with os.fdopen(fd, "w") as fh:
                    json.dump({"ident": ident, "version": _version}, fh)
                os.replace(tmp, memo)
            except OSError:
                pass
    return _version


def flags():
    # The reference compiles its wrappers with -O3 -ffast-math (pyop2/compilation.py:345-349).  Here: the
    # value-safe subset (no NaN/Inf/signed-zero bookkeeping, reassociation, contraction) so that 0*x and
    # 1*x fold in unrolled tabulation loops, but NOT approximate functions: f64 division keeps the
    # div_scale/div_fmas/div_fixup sequence (a bare v_rcp_f64 is not accurate to 1e-12).
    f = [f"--offload-arch={configuration['arch']}", "-O3", "-std=c++17", "--genco", "-munsafe-fp-atomics",
         "-fno-math-errno", "-fno-signed-zeros", "-fno-honor-nans", "-fno-honor-infinities", "-fassociative-math",
         "-fno-trapping-math", "-ffp-contract=fast", f"-I{_CSRC}"]
    if configuration["cflags"]:
        f += configuration["cflags"].split()
    return f


def _wrapper_header_hash():
    h = hashlib.sha1()
    for name in ("fd_wrapper.h", "fd_tensor.h", "fd_callables.h"):
        with open(os.path.join(_CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def compile_hip(source: str, name: str, extra_flags=()) -> str:
    """Return the path of the cached code object for ``source`` (compiling it if needed).  ``extra_flags``: further hipcc
    arguments for this kernel only (part of the cache key)."""
    cache = configuration["cache_dir"]
    os.makedirs(cache, exist_ok=True)
    fl = flags() + list(extra_flags)
    # every flag except the include path of this installation (the header's CONTENT is hashed instead)
    keyed = [f for f in fl if f != f"-I{_CSRC}"]
    key = hashlib.sha1("\0".join([source, " ".join(keyed), compiler_version(), _wrapper_header_hash()]).encode()).hexdigest()[:20]
    out = os.path.join(cache, f"{name}_{key}.hsaco")
    if os.path.exists(out):
        stats["cache_hits"] += 1
        try:
            os.utime(out)                 # "used now": evict_unused() removes code objects nothing has asked for in a while
        except OSError:
            pass
        return out
    stats["hipcc_runs"] += 1
    stats.setdefault("compiled", []).append(f"{name}_{key}")
    # the source goes to a unique temporary name and is renamed into place (pyop2/compilation.py:560-575): ranks that
    # miss the cache together never truncate a file another rank's hipcc is reading
    src = os.path.join(cache, f"{name}_{key}.hip")
    fd, tmpsrc = tempfile.mkstemp(suffix=".hip", dir=cache)
    with os.fdopen(fd, "w") as fh:
        fh.write(source)
    os.replace(tmpsrc, src)
    fd, tmp = tempfile.mkstemp(suffix=".hsaco", dir=cache)
    os.close(fd)
    # -Rpass-analysis=kernel-resource-usage: the register / scratch / occupancy figures of the wrapper kernel, kept in a
    # sidecar file next to the code object (kernel_resources) for the occupancy-directed variant choice in kernel.py
    cmd = [configuration["hipcc"], *fl, "-Rpass-analysis=kernel-resource-usage", "-o", tmp, src]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True)
    except OSError as e:
        os.unlink(tmp)
        raise CompilationError(f"cannot run {configuration['hipcc']}: {e}")
    if r.returncode != 0:
        os.unlink(tmp)
        raise CompilationError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stderr}")
    res = _parse_resources(r.stderr)
    if res:
        fd, rtmp = tempfile.mkstemp(suffix=".res.json", dir=cache)      # unique per process: ranks compiling the same
        with os.fdopen(fd, "w") as fh:                                  # kernel at the same time never share a temporary
            json.dump(res, fh)
        os.replace(rtmp, out + ".res.json")
    os.replace(tmp, out)      # atomic: concurrent ranks race benignly
    return out


_RES_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
             "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "static_lds"}


def _parse_resources(stderr: str):
    """{kernel name: {vgprs, scratch, occupancy, ...}} from hipcc's kernel-resource-usage remarks."""
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in _RES_KEYS:
            cur[_RES_KEYS[m.group(1).strip()]] = int(m.group(2))
    return out


def kernel_resources(path: str, symbol: str):
    """Resource usage of ``symbol`` in the cached code object ``path`` (None when it was compiled without remarks)."""
    try:
        with open(path + ".res.json") as fh:
            return json.load(fh).get(symbol)
    except (OSError, ValueError):
        return None


def evict_unused(max_age_hours=36.0, cache=None):
    """Remove the code objects (with their sources and resource sidecars) that no process has asked for within ``max_age_hours``:
    every change of fd_wrapper.h / fd_tensor.h / the compiler starts a new generation of keys, and the old generation can never be
    hit again.  The reference leaves its cache to the user (pyop2/compilation.py:424-455, ``firedrake-clean``); here the cache ships
    with the tree, so ``__graft_entry__.build()`` prunes it.  Returns (kept, removed) counts of code objects."""
    import time
    cache = cache or configuration["cache_dir"]
    try:
        names = os.listdir(cache)
    except OSError:
        return 0, 0
    limit = time.time() - max_age_hours * 3600.0
    live = {n[:-len(".hsaco")] for n in names if n.endswith(".hsaco") and os.path.getmtime(os.path.join(cache, n)) >= limit}
    kept = removed = 0
    for n in names:
        stem = n
        for suffix in (".hsaco.res.json", ".hsaco", ".hip"):
            if n.endswith(suffix):
                stem = n[:-len(suffix)]
                break
        else:
            continue
        if stem in live:
            kept += n.endswith(".hsaco")
            continue
        path = os.path.join(cache, n)
        if n.endswith(".hip") and os.path.getmtime(path) >= limit:
            continue                      # a compile in flight (source written, code object not yet renamed into place)
        try:
            os.unlink(path)
            removed += n.endswith(".hsaco")
        except OSError:
            pass
    return kept, removed
