#!/usr/bin/env python3
"""instruction mix of one kernel in a gfx950 .s file (hipcc -save-temps): usage isa_mix.py file.s substring-of-kernel-symbol"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if key in l and l.split(";")[0].strip().endswith(":") and not l.startswith(".") and not l.startswith("\t"))
cnt = collections.Counter(); n = 0
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end") or t.startswith("s_endpgm"): 
        if t.startswith("s_endpgm"): cnt["s_endpgm"] += 1
        if t.startswith(".Lfunc_end"): break
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    op = t.split()[0]
    n += 1
    if op.startswith("v_accvgpr"): cnt["accvgpr mov"] += 1
    elif op.startswith("scratch_"): cnt[op.split("_")[0] + "_" + op.split("_")[1]] += 1
    elif "dpp" in t and op.startswith("v_fmac_f64"): cnt["v_fmac_f64_dpp"] += 1
    elif op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): cnt["v_fma_f64"] += 1
    elif op.startswith("v_mul_f64"): cnt["v_mul_f64"] += 1
    elif op.startswith("v_add_f64"): cnt["v_add_f64"] += 1
    elif op.startswith("v_mov_b64") or op.startswith("v_mov_b32"): cnt[op.split("_e")[0]] += 1
    elif op.startswith("v_div") or op.startswith("v_rcp"): cnt["div/rcp"] += 1
    elif op.startswith("global_load"): cnt["global_load"] += 1
    elif op.startswith("global_store"): cnt["global_store"] += 1
    elif op.startswith("s_nop"): cnt["s_nop"] += 1
    elif op.startswith("s_waitcnt"): cnt["s_waitcnt"] += 1
    elif op.startswith("v_"): cnt["other valu"] += 1
    elif op.startswith("s_"): cnt["other salu"] += 1
    else: cnt["other"] += 1
print(lines[start][:120], "static instructions:", n)
for k, v in cnt.most_common(): print("  %-18s %6d  %5.1f%%" % (k, v, 100.0*v/n))
