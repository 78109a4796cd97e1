#!/usr/bin/env python3
"""Remove the timing-only / ablation preprocessor hooks from the shipped kernel sources (a minimal `unifdef -U`): every macro in
UNDEF is treated as undefined.  The hooks live on in git history (the commits named in profiles/README.md), which is where the
A/B tools (tools/abl_*.py) build their variants from.  python tools/strip_ablation_hooks.py file..."""
import re, sys
UNDEF = {"BAMD_ABLATE_QWRITE", "BAMD_ABLATE_LRELU", "BAMD_ABLATE_XLOAD", "BAMD_ABLATE_HALF_LOADS", "BAMD_ABLATE_DW", "BAMD_ABLATE_CHAIN",
         "BAMD_ABLATE_EPILOGUE", "BAMD_DMA_ABL", "BAMD_DMA_NOTAIL", "BAMD_DMA_NOBAR", "BAMD_WB_ABL", "BAMD_INFER_NOPF", "BAMD_DW_NOLOAD",
         "BAMD_BF16_SCALAR_MUL", "BAMD_BF16_SPLIT_BALANCED", "BAMD_WT_ABL_X", "BAMD_WT_ABL_LOSS", "BAMD_DW_LOOSE"}
def truth(cond):
    """value of an #if / #ifdef / #ifndef condition when every UNDEF macro is undefined; None = not about them"""
    m = re.match(r"#\s*ifdef\s+(\w+)", cond)
    if m: return False if m.group(1) in UNDEF else None
    m = re.match(r"#\s*ifndef\s+(\w+)", cond)
    if m: return True if m.group(1) in UNDEF else None
    m = re.match(r"#\s*if\s+(.*)", cond)
    if m:
        e = m.group(1).split("//")[0].strip()
        names = set(re.findall(r"defined\((\w+)\)", e))
        if names and names <= UNDEF:
            e2 = re.sub(r"defined\(\w+\)", "0", e)
            e2 = re.sub(r"\b(%s)\b" % "|".join(UNDEF), "0", e2).replace("&&", " and ").replace("||", " or ").replace("!", " not ")
            return bool(eval(e2))
    return None
for path in sys.argv[1:]:
    out, stack = [], []      # stack of (known truth or None, in_else)
    for line in open(path).read().split("\n"):
        st = line.strip()
        if re.match(r"#\s*if", st):
            t = truth(st)
            emit_parent = all(a is None or (a != e) for a, e in stack)      # parent regions currently emitted
            stack.append((t, False))
            if t is None and emit_parent: out.append(line)
            continue
        if re.match(r"#\s*else", st) and stack:
            t, _ = stack[-1]
            stack[-1] = (t, True)
            if t is None and all(a is None or (a != e) for a, e in stack[:-1]): out.append(line)
            continue
        if re.match(r"#\s*endif", st) and stack:
            t, _ = stack.pop()
            if t is None and all(a is None or (a != e) for a, e in stack): out.append(line)
            continue
        if all(a is None or (a != e) for a, e in stack): out.append(line)
    open(path, "w").write("\n".join(out))
    print(path, "stripped")
