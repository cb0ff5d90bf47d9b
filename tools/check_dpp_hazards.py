"""Check the inline-asm DPP sequences of ptz_chol.hip in the compiler's output (the hazard recogniser does not look
inside inline asm).  gfx9 rule: a VALU write of a VGPR needs two wait states before a DPP instruction reads that VGPR as its DPP
source (src0).  Usage: check_dpp_hazards.py file.s  -- prints the violations, exit code 1 if any."""
import re
import sys


def regs(tok):
    """v12 -> {12}; v[12:13] -> {12, 13}; anything else -> empty (modifiers like -v[..] / |v..| stripped)."""
    tok = tok.strip().lstrip("-").strip("|")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(path):
    bad = []
    hist = []  # (wait_states, written vgprs, text) of the instructions before, newest last
    n_dpp = 0
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            if t.endswith(":"):
                hist = []  # a label: other paths lead here, nothing is known
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if "_dpp" in op or "row_newbcast" in t or "quad_perm" in t or "row_shr" in t or "row_ror" in t or "row_shl" in t:
            n_dpp += 1
            src0 = regs(ops[1].split(" ")[0]) if len(ops) > 1 else set()
            ws = 0
            for w, wr, txt in reversed(hist):
                if ws >= 2:
                    break
                if wr & src0:
                    bad.append((ln, t, txt))
                    break
                ws += w
        if op == "s_nop":
            hist.append((int(ops[0], 0) + 1 if ops else 1, set(), t))
        elif op.startswith("v_") and not op.startswith("v_cmp") and ops:
            hist.append((1, regs(ops[0].split(" ")[0]), t))
        else:
            hist.append((1, set(), t))
        hist = hist[-6:]
    return n_dpp, bad


if __name__ == "__main__":
    n, bad = scan(sys.argv[1])
    print(f"{n} DPP instructions, {len(bad)} with their source written less than two wait states before")
    for ln, t, txt in bad[:20]:
        print(f"  line {ln}: {t}   <-   {txt}")
    sys.exit(1 if bad else 0)
