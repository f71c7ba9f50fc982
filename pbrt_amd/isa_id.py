"""Identity of a KERNEL of the built library: a hash of its machine code.

A committed counter profile (profiles/pmc_<workload>.json) prices one kernel -- the production instantiation of render_kernel that
the workload launches -- and must be withheld when that kernel changes, and only then.  Round 4 keyed profiles by a hash of every
source file under csrc/ (pbrt_hip_build_id), so an edit to the scene parser or a new sampler instantiation invalidated three
measurement sets of a 270-minute GPU budget.  Here the key is what the GPU executes: the bytes of the
kernel's function in the gfx950 code object (.text) plus its 64-byte kernel descriptor (register counts, LDS, launch bounds: what
decides occupancy).  Pure Python over the file -- clang offload bundle -> ELF64 -> symbol table --, no tool of the ROCm tree needed;
names are demangled with c++filt where it exists.

  python -m pbrt_amd.isa_id [libpbrt_hip.so] [regex]     prints id + demangled name of every matching kernel
"""
import hashlib
import os
import re
import struct
import subprocess

BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob, arch="gfx950"):
    """the device ELF images of `arch` inside a host library (uncompressed clang offload bundles, one per translation unit)"""
    out, at = [], 0
    while True:
        at = blob.find(BUNDLE_MAGIC, at)
        if at < 0:
            return out
        n, = struct.unpack_from("<Q", blob, at + 24)
        p = at + 32
        if n > 64:  # (the magic as a string constant somewhere else)
            at += 24
            continue
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode("ascii", "replace")
            p += 24 + tlen
            if arch in triple and size and blob[at + off:at + off + 4] == b"\x7fELF":
                out.append(blob[at + off:at + off + size])
        at += 24


def _elf_symbols(elf):
    """-> (sections {name: (addr, offset, size)}, [(name, type, shndx, value, size)]) of an ELF64 little-endian image"""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    sh = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    strtab = sh[shstrndx]
    def sname(o):
        e = elf.index(b"\0", strtab[4] + o)
        return elf[strtab[4] + o:e].decode()
    sections = {sname(s[0]): (s[3], s[4], s[5]) for s in sh}
    by_index = [(s[3], s[4], s[5]) for s in sh]
    syms = []
    for s in sh:
        if s[1] != 2:  # SHT_SYMTAB
            continue
        names = sh[s[6]]  # sh_link -> its string table
        for i in range(s[5] // 24):
            st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, s[4] + i * 24)
            e = elf.index(b"\0", names[4] + st_name)
            syms.append((elf[names[4] + st_name:e].decode(), st_info & 15, st_shndx, st_value, st_size))
    return sections, by_index, syms


def _demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True, timeout=60)
            out = r.stdout.split("\n")
            if len(out) >= len(names):
                return out[:len(names)]
        except (OSError, subprocess.SubprocessError):
            continue
    return list(names)


def kernel_ids(lib_path, arch="gfx950"):
    """{demangled kernel name: 16 hex digits} for every kernel (a FUNC symbol with a `<name>.kd` descriptor) of the library"""
    found = kernel_ids_by_symbol(lib_path, arch)
    names = sorted(found)
    return {d: found[n] for n, d in zip(names, _demangle(names))}


def kernel_ids_by_symbol(lib_path, arch="gfx950"):
    """{MANGLED kernel symbol: 16 hex digits}: pure file parsing, no child process (what a GPU-initialised process may call: bench.py)"""
    blob = open(lib_path, "rb").read()
    found = {}
    for elf in _code_objects(blob, arch):
        _, secs, syms = _elf_symbols(elf)
        kd = {n[:-3]: (shndx, value, size) for n, t, shndx, value, size in syms if n.endswith(".kd") and size}
        for n, t, shndx, value, size in syms:
            if t != 2 or n not in kd or not size or shndx >= len(secs):  # STT_FUNC with a descriptor
                continue
            addr, off, _ = secs[shndx]
            h = hashlib.sha256(elf[off + value - addr:off + value - addr + size])
            dx, dv, ds = kd[n]
            daddr, doff, _ = secs[dx]
            desc = bytearray(elf[doff + dv - daddr:doff + dv - daddr + ds])
            desc[16:24] = b"\0" * 8  # KERNEL_CODE_ENTRY_BYTE_OFFSET: where the code sits relative to the descriptor, not what it is
            desc[8:12] = b"\0" * 4   # KERNARG_SIZE: an argument appended for ANOTHER kernel's use grows it and changes nothing this one runs
            h.update(bytes(desc))
            found[n] = h.hexdigest()[:16]
    return found


def kernel_resources(lib_path, symbol, arch="gfx950"):
    """{vgpr_count, sgpr_count, agpr_count, vgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size (STATIC LDS; the render
    kernels' stack is dynamic), wavefront_size, max_flat_workgroup_size} of the kernel with mangled symbol `symbol`, from the code object's
    NT_AMDGPU_METADATA note (what the compiler allocated -- rocprofv3's VGPR_Count / LDS_Block_Size columns are granules and launch-time
    fields: VERDICT r05 weak 2c); None if the library has no such kernel.  File parsing + msgpack, no child process."""
    import msgpack
    blob = open(lib_path, "rb").read()
    for elf in _code_objects(blob, arch):
        shoff, = struct.unpack_from("<Q", elf, 0x28)
        shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
        for i in range(shnum):
            sh = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
            if sh[1] != 7:  # SHT_NOTE
                continue
            p, end = sh[4], sh[4] + sh[5]
            while p + 12 <= end:
                namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
                d0 = p + 12 + ((namesz + 3) & ~3)
                if ntype == 32:  # NT_AMDGPU_METADATA
                    meta = msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
                    for k in meta.get("amdhsa.kernels", []):
                        if k.get(".name") == symbol:
                            return {f: k.get("." + f) for f in ("vgpr_count", "sgpr_count", "agpr_count", "vgpr_spill_count", "private_segment_fixed_size",
                                                                 "group_segment_fixed_size", "wavefront_size", "max_flat_workgroup_size")}
                p = d0 + ((descsz + 3) & ~3)
    return None


def kernel_symbol(lib_path, kernel_name):
    """the mangled symbol of the kernel a profiler's demangled name stands for (runs the demangler: call it where no GPU is initialised,
    tools/summarize_profile.py), or None"""
    found = kernel_ids_by_symbol(lib_path)
    names = sorted(found)
    want = normalise(kernel_name)
    for n, d in zip(names, _demangle(names)):
        if normalise(d) == want:
            return n
    return None


def normalise(kernel_name):
    """`render_kernel<false, false, false, 30, 3, false, false>` out of whatever a profiler or demangler wrapped around it"""
    m = re.search(r"(\w+<[^()]*>)\s*\(", kernel_name + "(")
    return re.sub(r"\s+", "", m.group(1)) if m else kernel_name.strip()


_cache = {}


def kernel_id(lib_path, kernel_name, symbol=None):
    """the id of the kernel a profile names (rocprofv3's demangled kernel name), or None if the library has no such kernel.  With the
    profile's `kernel_symbol` (the mangled name, stored by tools/summarize_profile.py) the lookup is file parsing alone: no demangler, so
    no child process is started from a process that has initialised the GPU."""
    if symbol:
        skey = ("sym", os.path.realpath(lib_path), os.path.getmtime(lib_path), os.path.getsize(lib_path))
        if skey not in _cache:
            _cache[skey] = kernel_ids_by_symbol(lib_path)
        if symbol in _cache[skey]:
            return _cache[skey][symbol]
        return None
    key = (os.path.realpath(lib_path), os.path.getmtime(lib_path), os.path.getsize(lib_path))
    if key not in _cache:
        _cache[key] = {normalise(k): v for k, v in kernel_ids(lib_path).items()}
    return _cache[key].get(normalise(kernel_name))


if __name__ == "__main__":
    import sys
    from .build import LIB_PATH
    path = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else LIB_PATH
    pat = re.compile(sys.argv[-1] if len(sys.argv) > 1 and sys.argv[-1] != path else ".")
    for name, kid in sorted(kernel_ids(path).items()):
        if pat.search(name) or pat.search(normalise(name)):
            print(kid, normalise(name))
