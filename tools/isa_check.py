"""What the compiler made of the kernels' memory accesses: per kernel of the built library, the number of flat / global loads and
stores (a flat access in a software-pipelined loop makes hipcc wait for EVERY outstanding request).

    python tools/isa_check.py [path/to/libplayrender.so]          # table of the kernels that still have flat accesses

Reads the gfx950 code objects out of the library's .hip_fatbin section (uncompressed clang offload bundles) and disassembles them with
/opt/rocm/lib/llvm/bin/llvm-objdump.  Used by tests/test_cpu.py::test_product_kernels_have_no_flat_memory_operations."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(library_path, arch="gfx950"):
    """the device ELF images for `arch` inside the library (one per translation unit)"""
    blob = open(library_path, "rb").read()
    images = []
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (count,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        at = base + len(MAGIC) + 8
        for _ in range(count):
            offset, size, triple_len = struct.unpack_from("<QQQ", blob, at)
            triple = blob[at + 24:at + 24 + triple_len].decode()
            at += 24 + triple_len
            if arch in triple and size:
                images.append(blob[base + offset:base + offset + size])
    return images


def memory_operations(library_path, arch="gfx950"):
    """{kernel symbol: {"flat_load": n, "flat_store": n, "global_load": n, "global_store": n}} over the library's kernels"""
    counts = {}
    with tempfile.TemporaryDirectory() as tmp:
        for i, image in enumerate(code_objects(library_path, arch)):
            path = os.path.join(tmp, f"{i}.co")
            with open(path, "wb") as f:
                f.write(image)
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
            name = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    name = m.group(1)
                    counts.setdefault(name, {"flat_load": 0, "flat_store": 0, "global_load": 0, "global_store": 0})
                    continue
                if name is None:
                    continue
                t = line.split()
                if not t:
                    continue
                op = t[0]
                for kind in ("flat", "global"):
                    if op.startswith(kind + "_load"):
                        counts[name][kind + "_load"] += 1
                    elif op.startswith(kind + "_store") or op.startswith(kind + "_atomic"):
                        counts[name][kind + "_store"] += 1
    return counts


def matrix_loops(library_path, arch="gfx950", min_mfma=8, max_instructions=600):
    """The innermost loops that hold matrix instructions (the K loops), per kernel:
    {kernel symbol: [{"mfma": n, "instructions": n, "waits": ["vmcnt(3) lgkmcnt(3)", ...]}, ...]}.
    A loop = a backward branch and the instructions between its target and itself."""
    loops = {}
    with tempfile.TemporaryDirectory() as tmp:
        for i, image in enumerate(code_objects(library_path, arch)):
            path = os.path.join(tmp, f"{i}.co")
            with open(path, "wb") as f:
                f.write(image)
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
            functions, name = {}, None
            for line in text.splitlines():
                m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
                if m:
                    name = m.group(2)
                    functions[name] = {"start": int(m.group(1), 16), "code": []}
                    continue
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-Fa-f]+):", line)
                if m and name is not None:
                    functions[name]["code"].append((int(m.group(3), 16), m.group(1), m.group(2), line))
            for name, fn in functions.items():
                code = fn["code"]
                index = {addr: k for k, (addr, _, _, _) in enumerate(code)}
                spans = []
                for k, (addr, op, _, line) in enumerate(code):
                    if op.startswith("s_cbranch") or op == "s_branch":
                        m = re.search(r"<[^>]+\+0x([0-9a-f]+)>", line)
                        if m:
                            target = fn["start"] + int(m.group(1), 16)
                            if target <= addr and target in index:
                                spans.append((index[target], k))
                found = []
                for (a, b) in spans:
                    body = code[a:b + 1]
                    n_mfma = sum(1 for (_, op, _, _) in body if op.startswith("v_mfma"))
                    if n_mfma < min_mfma or len(body) > max_instructions:
                        continue
                    # not innermost: a shorter loop with matrix instructions starts inside this one (nested, or - a rotated layer loop
                    # around a K loop - overlapping: the K loop's back edge then lies behind the layer loop's)
                    if any((a2, b2) != (a, b) and a <= a2 <= b and (b2 - a2) < (b - a) and
                           sum(1 for (_, op, _, _) in code[a2:b2 + 1] if op.startswith("v_mfma")) >= min_mfma for (a2, b2) in spans):
                        continue
                    found.append({"mfma": n_mfma, "instructions": len(body),
                                  "waits": [args for (_, op, args, _) in body if op == "s_waitcnt"]})
                if found:
                    loops[name] = found
    return loops


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "playableenvironments_amd", "libplayrender.so")
    counts = memory_operations(path)
    flat = {k: v for k, v in counts.items() if v["flat_load"] or v["flat_store"]}
    print(f"{len(counts)} kernels, {len(flat)} with flat accesses")
    for k, v in sorted(flat.items()):
        print(f"  {k[:90]:90s} {v}")
    print("K loops whose waits include vmcnt(0) (every outstanding request):")
    for k, found in sorted(matrix_loops(path).items()):
        if "gemm" in k:
            continue        # (the LDS-staged GEMMs wait for a whole slab in front of its staging: by design)
        for loop in found:
            if any("vmcnt(0)" in w for w in loop["waits"]):
                print(f"  {k[:70]:70s} {loop['mfma']:3d} MFMAs / {loop['instructions']:4d} instructions: {' | '.join(loop['waits'])[:120]}")


if __name__ == "__main__":
    main()
