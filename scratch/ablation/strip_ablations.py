#!/usr/bin/env python3
"""Removes the timing-ablation / A-B preprocessor branches (macros that no shipped build defines) from the MLP kernel sources, keeping the branch a shipped build compiles.
A partial `unifdef`: conditions are flat expressions of defined(X) / !defined(X) joined by && or ||, plus atoms it does not understand (kept as they are).
    python scratch/ablation/strip_ablations.py <in> <out>
The kernels WITH their ablation switches (what scratch/*_ab.sh, build_ablate.sh, mxdev.sh, prologue_ab.sh, trunk_ab.sh compile) are kept beside this script."""
import re
import sys

UNDEF = {"IBL_ABLATE_LOOPONLY", "IBL_ABLATE_NO_BARRIER", "IBL_ABLATE_NO_EPI", "IBL_ABLATE_NO_FRAG", "IBL_ABLATE_NO_LOADS", "IBL_ABLATE_NO_MFMA", "IBL_ABLATE_RELU_BITS",
         "IBL_MX_ABLATE_HALF_LOADS", "IBL_MX_ABLATE_NO_BARRIER", "IBL_MX_ABLATE_NO_EPI", "IBL_MX_ABLATE_NO_HEADS", "IBL_MX_ABLATE_NO_LOADS", "IBL_MX_ABLATE_PROLOGUE",
         "IBL_MX_AGPR_CHAIN", "IBL_MX_DEV_TRUNK_ONLY", "IBL_MX_DOUBLE_DMA", "IBL_MX_DOUBLE_LDS", "IBL_MX_DOUBLE_MFMA", "IBL_MX_HEADS_PK_FMA", "IBL_MX_NO_EST", "IBL_NO_POINT_GEN",
         "IBL_TRACE", "IBL_DUAL_ACC"}


def simplify(expr):
    """-> True | False | a rewritten expression string"""
    expr = expr.split("//")[0].strip()
    for op in ("||", "&&"):
        other = "&&" if op == "||" else "||"
        if op in expr and other not in expr:
            terms = [t.strip() for t in expr.split(op)]
            break
    else:
        if "||" in expr and "&&" in expr:
            if not any(m in expr for m in UNDEF):
                return expr
            raise SystemExit("mixed && / || with an ablation macro: " + expr)
        terms, op = [expr], "&&"
    out = []
    for t in terms:
        m = re.fullmatch(r"(!?)\s*defined\s*\(?\s*(\w+)\s*\)?", t)
        if m and m.group(2) in UNDEF:
            v = bool(m.group(1))            # !defined(U) -> True, defined(U) -> False
            if op == "||" and v:
                return True
            if op == "&&" and not v:
                return False
            continue
        if any(u in t for u in UNDEF):
            raise SystemExit("cannot simplify term: " + t)
        out.append(t)
    if not out:
        return op == "&&"
    return (" %s " % op).join(out)


def main(src, dst):
    lines = open(src).read().split("\n")
    out = []
    # stack entries: dict(kind: 'keep' (condition untouched / rewritten, directives stay) | 'resolved' (directives dropped), emitting: bool, done: bool (a branch was taken))
    stack = []

    def emitting():
        return all(f["emitting"] for f in stack)

    for ln in lines:
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", ln)
        if not m:
            if emitting():
                out.append(ln)
            continue
        d, rest = m.group(1), m.group(2)
        if d in ("ifdef", "ifndef", "if"):
            cond = ("defined(%s)" % rest.split()[0]) if d == "ifdef" else ("!defined(%s)" % rest.split()[0]) if d == "ifndef" else rest
            comment = ("  //" + rest.split("//", 1)[1]) if "//" in rest and d != "if" else ""
            s = simplify(cond) if emitting() else cond
            if not emitting():
                stack.append({"kind": "dead", "emitting": False, "done": True})
            elif s is True or s is False:
                stack.append({"kind": "resolved", "emitting": s, "done": s})
            else:
                stack.append({"kind": "keep", "emitting": True, "done": False, "opened": True})
                out.append(ln if s == cond.split("//")[0].strip() else "#if " + s + comment)
        elif d == "elif":
            f = stack[-1]
            if f["kind"] == "dead":
                continue
            parent_emitting = all(g["emitting"] for g in stack[:-1])
            s = simplify(rest) if parent_emitting else rest
            if f["kind"] == "resolved":
                if f["done"]:
                    f["emitting"] = False
                elif s is True:
                    f["emitting"], f["done"] = True, True
                elif s is False:
                    f["emitting"] = False
                else:      # the chain's first live condition: it opens a kept chain
                    f.update(kind="keep", emitting=True, opened=True)
                    out.append("#if " + s)
            else:      # keep
                if s is True:
                    out.append("#else")
                    f["swallow_rest"] = True
                elif s is False or f.get("swallow_rest"):
                    f["emitting"] = False if s is False else f["emitting"]
                    if s is False:
                        f["skip_branch"] = True
                        f["emitting"] = False
                else:
                    f["emitting"] = True
                    out.append("#elif " + s)
        elif d == "else":
            f = stack[-1]
            if f["kind"] == "dead":
                continue
            if f["kind"] == "resolved":
                f["emitting"] = not f["done"]
                f["done"] = True
            else:
                if f.get("swallow_rest"):
                    f["emitting"] = False
                else:
                    f["emitting"] = True
                    out.append(ln)
        else:      # endif
            f = stack.pop()
            if f["kind"] == "keep":
                out.append(ln)
    assert not stack
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
