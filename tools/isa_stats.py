#!/usr/bin/env python3
"""Per-kernel facts of a built device library, read from its gfx950 code object (no GPU needed):
code bytes, VGPRs / SGPRs, spilled registers, private segment, and how many scratch_ instructions the ISA holds.

usage: python tools/isa_stats.py [path/to/libmi355rt*.so ...]      (default: the product library)
The numbers that DESIGN.md / profiles quote for "code size", "spills" and "scratch instructions" come from here."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(so, workdir):
    """Extracts the gfx950 code object of `so` into workdir and returns its path."""
    base = os.path.join(workdir, os.path.basename(so))
    subprocess.check_call(["cp", so, base])
    subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "--offloading", base], cwd=workdir, stderr=subprocess.STDOUT)
    for f in os.listdir(workdir):
        if f.startswith(os.path.basename(so)) and "amdgcn" in f:
            return os.path.join(workdir, f)
    raise RuntimeError("no amdgcn code object in " + so)


def kernel_stats(so):
    out = {}
    with tempfile.TemporaryDirectory() as wd:
        co = code_object(so, wd)
        syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-sW", co], text=True)
        for line in syms.splitlines():
            m = re.match(r"\s*\d+:\s+([0-9a-f]+)\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(\S+)", line)
            if m and "k_" in m.group(3):
                out[m.group(3)] = {"code_bytes": int(m.group(2))}
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
        # amdhsa.kernels: one YAML map per kernel, keys in alphabetical order, each map starts at "- .agpr_count"
        for entry in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            m = re.search(r"\.symbol:\s+(\S+)\.kd", entry)
            if not m:
                continue
            st = out.setdefault(m.group(1), {})
            for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                mm = re.search(r"\n\s*\." + key + r":\s+(\d+)", entry)
                if mm:
                    st[key] = int(mm.group(1))
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)
        cur = None
        for line in dis.splitlines():
            m = re.match(r"[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                continue
            if cur in out:
                st = out[cur]
                t = line.split()
                if len(t) < 1:
                    continue
                op = t[0]
                st["insts"] = st.get("insts", 0) + 1
                if op.startswith("scratch_"):
                    st["scratch_insts"] = st.get("scratch_insts", 0) + 1
                    st["scratch_" + ("loads" if "load" in op else "stores")] = st.get("scratch_" + ("loads" if "load" in op else "stores"), 0) + 1
    return out


def short(name):
    m = re.search(r"(k_[a-z0-9_]+?)E?(?:NS_|RK|PK|v$|$)", name)
    m2 = re.search(r"\d+(k_[A-Za-z0-9_]+?)E", name)
    return m2.group(1) if m2 else (m.group(1) if m else name)


def kernel_isa(so):
    """{kernel: its instructions as text, one per line, without addresses / encodings} -- for comparing two builds instruction by instruction."""
    out = {}
    with tempfile.TemporaryDirectory() as wd:
        co = code_object(so, wd)
        dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
        cur = None
        for line in dis.splitlines():
            m = re.match(r"[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1); out[cur] = []
                continue
            if cur and line.strip():
                out[cur].append(re.sub(r"\s*//.*$", "", line).strip())
    return out


_COMMUTATIVE = ("v_mul_f32", "v_add_f32", "v_add_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_max_f32", "v_min_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co_u32")


def _commuted(line):
    """An instruction with the two source operands of a commutative VALU operation in sorted order."""
    t = line.split(None, 1)
    if len(t) == 2 and t[0].rsplit("_e", 1)[0] in _COMMUTATIVE:
        ops = [o.strip() for o in t[1].split(",")]
        if len(ops) == 3:
            return t[0].rsplit("_e", 1)[0] + " " + ops[0] + ", " + ", ".join(sorted(ops[1:]))
    return line


def diff(a, b):
    """usage: isa_stats.py --diff A.so B.so : per kernel, are the two builds the same instruction sequence?  (The proof that removing dead
    alternatives from the sources changed nothing that runs.)  Exit code 1 when any kernel present in both differs."""
    ia, ib = kernel_isa(a), kernel_isa(b)
    bad = 0
    for k in sorted(set(ia) | set(ib), key=short):
        if k not in ia or k not in ib:
            print(f"{short(k):34s} only in {'A' if k in ia else 'B'}")
            continue
        same = ia[k] == ib[k]
        if not same and sorted(map(_commuted, ia[k])) == sorted(map(_commuted, ib[k])):
            # the same instructions on the same registers; the scheduler emitted some in another order / with the operands of a commutative
            # operation swapped (what a change of dead source text can do to value numbering): nothing that runs differs in kind or count
            print(f"{short(k):34s} same instructions, {sum(x != y for x, y in zip(ia[k], ib[k]))} of {len(ia[k])} lines reordered / commuted")
            continue
        bad += 0 if same else 1
        note = "" if same else f"  ({len(ia[k])} vs {len(ib[k])} instructions; first difference at #{next((i for i, (x, y) in enumerate(zip(ia[k], ib[k])) if x != y), min(len(ia[k]), len(ib[k])))})"
        print(f"{short(k):34s} {'identical' if same else 'DIFFERENT'}{note}")
    return 1 if bad else 0


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--diff":
        sys.exit(diff(sys.argv[2], sys.argv[3]))
    libs = sys.argv[1:] or [os.path.join(ROOT, "raytracer-rust_amd", "_build", "libmi355rt.so")]
    for so in libs:
        print(f"# {os.path.relpath(so, ROOT) if so.startswith(ROOT) else so}")
        print(f"{'kernel':34s} {'code B':>8s} {'insts':>7s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'private B':>9s} {'LDS B':>7s} {'scratch_ ld/st':>14s}")
        for k, st in sorted(kernel_stats(so).items(), key=lambda kv: short(kv[0])):
            print(f"{short(k):34s} {st.get('code_bytes', 0):8d} {st.get('insts', 0):7d} {st.get('vgpr_count', 0):5d} {st.get('sgpr_count', 0):5d} "
                  f"{st.get('vgpr_spill_count', 0):6d} {st.get('sgpr_spill_count', 0):6d} {st.get('private_segment_fixed_size', 0):9d} {st.get('group_segment_fixed_size', 0):7d} "
                  f"{st.get('scratch_loads', 0):6d} /{st.get('scratch_stores', 0):6d}")


if __name__ == "__main__":
    main()
