"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py file.s SYMBOL_SUBSTRING [top]
(static counts per mnemonic and per class - a first look at what a vector-instruction-bound kernel spends its issue slots on)."""
import collections
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sym in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\ts_endpgm") or lines[i].startswith(".Lfunc_end"))
    ops = collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if m and not l.startswith("\t."):
            ops[m.group(1)] += 1
    cls = collections.Counter()
    for op, n in ops.items():
        if op.startswith("v_mfma"): c = "mfma"
        elif op.startswith("v_pk_"): c = "valu packed"
        elif op.startswith("v_cvt"): c = "valu cvt"
        elif op.startswith("v_"): c = "valu"
        elif op.startswith("ds_"): c = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
        elif op.startswith("s_"): c = "salu"
        else: c = "other"
        cls[c] += n
    print(lines[start][:120])
    print(dict(cls), "total", sum(ops.values()))
    for op, n in ops.most_common(top):
        print(f"  {n:6d} {op}")


if __name__ == "__main__":
    main()
