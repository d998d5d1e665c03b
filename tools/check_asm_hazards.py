"""Static check of generated gfx950 assembly: a scalar register written by the VECTOR unit (v_readfirstlane / v_readlane) must
not be used as the address base of a global_* instruction within 5 wait states.  hipcc's hazard recogniser places the s_nop
itself for instructions it emits, but it does not look inside inline asm -- and the visit loop of k_scatter_wide issues all its
VMEM through inline asm with SGPR bases (a no-compute ablation build whose bases came from v_readfirstlane faulted).
usage: python tools/check_asm_hazards.py <file.s> ...   (exit code 1 if a hazard is found)"""
import re
import sys


def scan(path):
    lines = [l.strip() for l in open(path)]
    lines = [l for l in lines if l and not l.startswith(";") and not l.startswith(".") and not l.endswith(":")]
    bad = []
    for i, l in enumerate(lines):
        m = re.match(r"global_(load|store|atomic)\S* .*s\[(\d+):(\d+)\]", l)
        if not m:
            continue
        lo, hi = int(m.group(2)), int(m.group(3))
        states, j = 0, i - 1
        while j >= 0 and states < 5:
            p = lines[j]
            w = re.match(r"v_read(first)?lane_b32 s(\d+),", p)
            if w and lo <= int(w.group(2)) <= hi:
                bad.append((p, l))
            k = re.match(r"s_nop (\d+)", p)
            states += int(k.group(1)) + 1 if k else 1
            j -= 1
    return bad


def reserved_sgpr_uses(path, lo=68, hi=99):
    """k_scatter_wide keeps its scalar entry stream in two FIXED SGPR tuples, s[68:83] and s[84:99], across inline-asm statements;
    the kernel is compiled with amdgpu_num_sgpr(76) so that hipcc itself stays inside s0..s67.  Returns every compiler-emitted
    instruction (outside ;;#ASMSTART .. ;;#ASMEND) of a k_scatter_wide body that names a register of that range."""
    bad, in_asm, in_kernel = [], False, False
    for raw in open(path):
        l = raw.strip()
        if l.endswith(":") or " ; @" in l:
            if re.match(r"_ZN\S*k_scatter_wide\S*:", l):
                in_kernel = True
            continue
        if l.startswith(".Lfunc_end"):
            in_kernel = False
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not in_kernel or in_asm or not l or l[0] in ";.":
            continue
        code = l.split(";")[0]
        for m in re.finditer(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]", code):
            a, b = (int(m.group(1)),) * 2 if m.group(1) else (int(m.group(2)), int(m.group(3)))
            if a <= hi and b >= lo:
                bad.append(l)
                break
    return bad


if __name__ == "__main__":
    total = 0
    for f in sys.argv[1:]:
        b = scan(f)
        for p, l in b:
            print(f"{f}: HAZARD {p}  ->  {l}")
        total += len(b)
        r = reserved_sgpr_uses(f)
        for l in r:
            print(f"{f}: RESERVED SGPR outside inline asm: {l}")
        total += len(r)
    print("hazards:", total)
    sys.exit(1 if total else 0)
