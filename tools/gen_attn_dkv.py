"""Writes vit_amd/csrc/attn_dkv64_body.inc: the instruction order of one step of the software-pipelined dK/dV pass.

One step of the key-stationary attention backward for head dim 64 works on three (32-query, 32-key) blocks at once:
  the 10 MFMAs that produce the scores S and dP of the NEXT block (two chains of 1 + 4),
  the 8 MFMAs that add the PREVIOUS block into dV and dK (its packed P / dS were made in the previous step),
and in the 18 gaps between them the vector work of the CURRENT block (exp2, multiply, convert/pack of 16 + 16 values)
and the LDS reads of later query blocks (row fragments: 9 x ds_read_b128, transposed fragments: 8 x 2
ds_read_b64_tr_b16).  Scores, packed operands and both fragment sets are double-buffered, so nothing a step's MFMAs
consume is produced inside that step: the matrix pipe never waits for a block's own exponentials or reads, and the
fillers can sit wherever the issue budget of a gap allows.

A wave issues in order, so WHERE each instruction sits between the MFMAs is the schedule; this script is the one place
that decides it.  The names (MF_*, EXP, MUL, CVP, CVD, PACK, RD_*, LD_*) are macros of attn.hip; every one of them expands
to volatile asm or to loads fenced by memory clobbers, so the emitted order is the order in the binary.

Placement rules (gfx950, one wave per SIMD):
  * a gap's fillers should issue within the MFMA's 32 cycles: v_exp_f32 costs 8 issue cycles, other vector instructions 4,
    an LDS read about 4, the MFMA itself holds the port for 8 -> BUDGET issue cycles of fillers per gap, at most
    MAX_EXP exponentials;
  * a vector instruction reads an MFMA result no sooner than three MFMAs after the one that wrote it (the hardware does
    not interlock that read): S / dP of the current block were finished a whole second half ago;
  * an instruction that consumes a vector result sits in a LATER gap than its producer (covers the one-instruction
    distance v_exp_f32 needs, and the convert -> MFMA distance);
"""
import os
import re
import sys

OUT = os.environ.get("DKV_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit_amd", "csrc",
                                               "attn_dkv64_body.inc")
# Timing-only ablations for the diagnostic library (results are wrong): DKV_ABLATE = comma list of
#   valu (no vector instructions), tread / aread (no transposed / row LDS reads), g1 / g2 (no MFMAs of that half),
#   stage (no global -> LDS loads inside the steps)
ABLATE = set(filter(None, os.environ.get("DKV_ABLATE", "").split(",")))
BUDGET = int(os.environ.get("DKV_BUDGET", "24"))
MAX_EXP = int(os.environ.get("DKV_MAX_EXP", "3"))
COST = {"EXP": 8, "MUL": 4, "CVP": 4, "CVD": 4}


def schedule_valu(n_gaps, reserved):
    """Greedy list schedule of the 48 vector instructions over the gaps.  reserved[g] = issue cycles already taken in gap g
    (LDS reads, loads).  Returns per-gap lists and the gap in which each pack completes."""
    ops = []                                   # (name, text, deps)
    for i in range(16):
        ops.append((f"E{i}", f"EXP({i});", []))
    for i in range(16):
        ops.append((f"M{i}", f"MUL({i});", [f"E{i}"]))
    for sx in range(2):
        for q in range(4):
            a, b = 8 * sx + 2 * q, 8 * sx + 2 * q + 1
            ops.append((f"P{sx}{q}", f"CVP({sx}, {q});", [f"E{a}", f"E{b}"]))
            ops.append((f"D{sx}{q}", f"CVD({sx}, {q});", [f"M{a}", f"M{b}"]))
    # priority: by the MFMA that first consumes the result (packed P of a query sub-block is needed two MFMAs before its
    # packed dS), then by value index, producers before consumers
    def prio(op):
        name = op[0]
        kind = name[0]
        idx = int(name[1:]) if kind in "EM" else 8 * int(name[1]) + 2 * int(name[2]) + 1
        sx = idx // 8
        deadline = 4 * sx + (0 if kind in "EP" else 2)
        return (deadline, idx // 2, "EMPD".index(kind))
    ops.sort(key=prio)
    placed, gaps = {}, [[] for _ in range(n_gaps)]
    for g in range(n_gaps):
        used, exps = reserved[g], 0
        for name, text, deps in ops:
            if name in placed:
                continue
            if any(d not in placed or placed[d] >= g for d in deps):
                continue
            kind = {"E": "EXP", "M": "MUL", "P": "CVP", "D": "CVD"}[name[0]]
            if used + COST[kind] > BUDGET or (kind == "EXP" and exps >= MAX_EXP):
                continue
            placed[name] = g
            gaps[g].append(text)
            used += COST[kind]
            exps += kind == "EXP"
    assert len(placed) == len(ops), f"vector work does not fit: {len(placed)} of {len(ops)} placed (raise DKV_BUDGET)"
    done = {(k, sx): max(placed[f"{k}{sx}{q}"] for q in range(4)) for k in "PD" for sx in range(2)}
    return gaps, done


def body(kb):
    """Instruction order of the step for key sub-block `kb` (the two differ in which LDS reads they carry)."""
    first = ["MF_S0();"] + [f"MF_S({k});" for k in range(4)] + ["MF_P0();"] + [f"MF_P({k});" for k in range(4)]
    second = [f"MF_{w}({sx}, {cb});" for sx in range(2) for w in ("DV", "DK") for cb in range(2)]
    mfmas = first + second
    n = len(mfmas)
    t_reads = [f"RD_{w}({sx}, {cb});" for sx in range(2) for w in ("DOT", "QT") for cb in range(2)]      # 8 x 2 instructions
    a_reads = ["RD_REC();"] + [f"RD_Q({k});" for k in range(4)] + [f"RD_DO({k});" for k in range(4)]     # 9 instructions
    # LDS reads are not cheap to issue (measured: about 8 cycles per ds_read_b128 and 12 per pair of transposed reads),
    # and a gap whose fillers exceed its budget delays the next MFMA by the excess while a gap below it gives nothing
    # back.  So the reads go one per gap into the second half of the step and the vector work, which has no deadline
    # inside the step (its packed results feed the NEXT step's dV / dK MFMAs), fills the gaps from the start up to the
    # budget, leaving room for the read where there is one:
    #   first key sub-block's step   the 9 row reads of the next query block; the next step's MFMAs use them in the order
    #                                they are read, each nine gaps after its read;
    #   second key sub-block's step  the 8 transposed fragments of the next query block (used a step later).
    # One of the tile's global -> LDS loads goes into every step (LD_STEP), alone in its gap.
    if kb == 0:
        r_gap = {n - len(a_reads) + i: r for i, r in enumerate(a_reads)}
        cost = 8
    else:
        r_gap = {n - len(t_reads) + i: r for i, r in enumerate(t_reads)}
        cost = 12
    l_gap = min(r_gap) - 1
    reserved = [cost if g in r_gap else (BUDGET if g == l_gap else 0) for g in range(n)]
    valu, done = schedule_valu(n, reserved)
    # The workgroup barrier of a tile sits inside its last step, behind that step's vector work: vector instructions issued
    # in the first few hundred cycles after a barrier release pay a start-up penalty, MFMAs and LDS reads do not, and
    # behind this point the step has only those left.
    sync_gap = max(g for g in range(n) if valu[g])
    lines = [f"// key sub-block {kb}: vector instructions per gap {[len(v) for v in valu]}; packs complete in gaps {done}"]
    for g, mf in enumerate(mfmas):
        if g == len(first):
            lines.append("// ---- dV / dK of the PREVIOUS block (packed P / dS from the previous step)")
        lines.append(mf)
        if g in r_gap:
            lines.append(r_gap[g])
        if g == l_gap:
            lines.append("LD_STEP();")
        if valu[g]:
            lines.append(" ".join(valu[g]))
        for (k, sx), dg in done.items():
            if dg == g:
                lines.append(f"PACK{k}({sx});")
        if kb == 1 and g == sync_gap:
            lines.append("if constexpr (QB == 1) { TILE_SYNC(); }")
    return lines


def main():
    lines = ["// generated by tools/gen_attn_dkv.py -- do not edit; change the generator and run it again",
             f"// fillers per gap: budget {BUDGET} issue cycles, at most {MAX_EXP} v_exp_f32",
             "if constexpr (KB == 0) {"] + body(0) + ["} else {"] + body(1) + ["}"]
    if ABLATE:
        drop = []
        if "valu" in ABLATE: drop += ["EXP(", "MUL(", "CVP(", "CVD("]
        if "tread" in ABLATE: drop += ["RD_DOT(", "RD_QT("]
        if "aread" in ABLATE: drop += ["RD_REC(", "RD_Q(", "RD_DO("]
        if "g1" in ABLATE: drop += ["MF_S0(", "MF_S(", "MF_P0(", "MF_P("]
        if "g2" in ABLATE: drop += ["MF_DV(", "MF_DK("]
        if "stage" in ABLATE: drop += ["LD_STEP("]
        pat = re.compile("|".join(re.escape(d) + r"[^;]*;" for d in drop))
        lines = [pat.sub("", ln) for ln in lines]
    with open(OUT, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"wrote {OUT}")


if __name__ == "__main__":
    sys.exit(main())
