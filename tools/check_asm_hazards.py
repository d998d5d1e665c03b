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


def _regs(tok):
    """vector registers named by one operand token: v7 -> {7}, v[4:7] -> {4..7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def inflight_register_uses(path, kernel=r"k_scatter_wide"):
    """The hazard class found in round 5: k_scatter_wide keeps asm-issued loads (global_load_*, ds_*_rtn / ds_read_*) IN FLIGHT
    across compiler-visible code; the landing register is tied to the asm statement that waits for it.  A compiler-emitted
    instruction (a v_mov copy resolving a phi, a spill, ...) that names such a register between the issuing statement and the
    statement that waits reads a value that has not landed.  Linear scan of the kernel body: landing registers of loads issued
    inside ;;#ASMSTART..;;#ASMEND are in flight until the next s_waitcnt of their counter (vmcnt / lgkmcnt; ANY count clears the
    whole class -- counted waits retire the oldest loads only, so this under-reports and never flags a landed register);
    returns the compiler-emitted instructions that name an in-flight register."""
    bad, in_asm, in_kernel = [], False, False
    flight = {"vm": set(), "lgkm": set()}
    for raw in open(path):
        l = raw.strip()
        if l.endswith(":") or " ; @" in l:
            if re.match(r"_ZN\S*" + kernel + r"\S*:", l):
                in_kernel, flight = True, {"vm": set(), "lgkm": set()}
            continue
        if l.startswith(".Lfunc_end"):
            in_kernel = False
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not in_kernel or not l or l[0] in ";.":
            continue
        code = l.split(";")[0].strip()
        if code.startswith("s_waitcnt"):
            if "vmcnt" in code:
                flight["vm"].clear()
            if "lgkmcnt" in code:
                flight["lgkm"].clear()
            if re.fullmatch(r"s_waitcnt\s+(0|0x0)", code):
                flight["vm"].clear(), flight["lgkm"].clear()
            continue
        ops = [t.strip() for t in re.split(r"[,\s]+", code)[1:]]
        named = set().union(*[_regs(t) for t in ops]) if ops else set()
        if in_asm:
            mnem = code.split()[0]
            if mnem.startswith("global_load") and ops:
                flight["vm"] |= _regs(ops[0])
            elif (mnem.startswith("ds_read") or "_rtn_" in mnem) and ops:
                flight["lgkm"] |= _regs(ops[0])
            continue
        hit = named & (flight["vm"] | flight["lgkm"])
        if hit:
            bad.append(l)
    return bad


def wide_kernel_facts(path):
    """(next_free_sgpr values, {instantiation: num_vgpr}) of the k_scatter_wide kernels in the file."""
    txt = open(path).read()
    sg = set()
    for m in re.finditer(r"\.amdhsa_kernel (\S*k_scatter_wide\S*)(.*?)\.end_amdhsa_kernel", txt, flags=re.S):
        sg |= set(re.findall(r"\.amdhsa_next_free_sgpr (\d+)", m.group(2)))
    vg = dict(re.findall(r"k_scatter_wideILb([01])E\S*\.num_vgpr, (\d+)", txt))
    return sg, {k: int(v) for k, v in vg.items()}


def check_wide(path):
    """Everything the build asserts about a scatter_wide assembly (Makefile: every variant; tests/test_capi_cpu.py: -O3, -O2)."""
    msgs = [f"HAZARD {p}  ->  {l}" for p, l in scan(path)]
    msgs += [f"RESERVED SGPR outside inline asm: {l}" for l in reserved_sgpr_uses(path)]
    msgs += [f"IN-FLIGHT landing register named by compiler code: {l}" for l in inflight_register_uses(path)]
    sg, vg = wide_kernel_facts(path)
    if sg != {"100"}:
        msgs.append(f"k_scatter_wide: .amdhsa_next_free_sgpr must be 100 (the descriptor must cover s[68:99]), found {sorted(sg)}")
    if len(vg) != 2:
        msgs.append(f"k_scatter_wide: expected two instantiations, found {vg}")
    txt = open(path).read()
    for t in ("s_load_dwordx16 s[68:68+15]", "s_load_dwordx16 s[84:84+15]"):
        if t not in txt:
            msgs.append(f"k_scatter_wide: the scalar entry stream's {t} is missing")
    return msgs


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--wide":  # the Makefile's build-time gate for every scatter_wide object
        msgs = [f"{f}: {m}" for f in sys.argv[2:] for m in check_wide(f)]
        print("\n".join(msgs) if msgs else "scatter_wide asm checks ok: " + " ".join(sys.argv[2:]))
        sys.exit(1 if msgs else 0)
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
