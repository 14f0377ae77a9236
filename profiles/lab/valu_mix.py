#!/usr/bin/env python
"""Weights every kernel's VALU instruction mix by the issue rates MEASURED on the box (profiles/r04_issue_rates.jsonl, from
profiles/lab/issue_rates.hip) -> profiles/valu_mix.json: cycles per wave64 VALU instruction, per kernel.

    python profiles/lab/valu_mix.py            (no GPU needed: disassembles tsfresh_amd/csrc/tsfa_kernels.o)

The mix is the STATIC histogram of the kernel's code object (llvm-objdump), used as a proxy for the dynamic mix the PMC
counter SQ_INSTS_VALU counts: the hot loops of these kernels are straight-line and unrolled, so the proxy is close, but it
is a proxy and the json says so.  `ms at full issue` of a kernel = SQ_INSTS_VALU x (weighted cycles per instruction) /
(1024 SIMDs x measured shader clock)."""
import collections
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OBJ = os.path.join(ROOT, "tsfresh_amd", "csrc", "tsfa_kernels.o")
LLVM = "/opt/rocm/lib/llvm/bin"


def rates():
    """base opcode (v_xxx without encoding suffix) -> cycles per wave-instruction per SIMD at 8 wavefronts per SIMD"""
    r, ghz = {}, []
    for line in open(os.path.join(ROOT, "profiles", "r04_issue_rates.jsonl")):
        d = json.loads(line)
        if d["waves_per_simd"] != 8:
            continue
        r[d["inst"]] = d["cycles_per_inst"]
        ghz.append(d["shader_ghz"])
    return r, sum(ghz) / len(ghz)


def classify(op, R):
    """-> (cycles, how) for one disassembled VALU mnemonic"""
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    dpp = op.endswith("_dpp") or op.endswith("_sdwa")
    name = base[2:]
    if dpp:
        return R["and_b32_dpp_row_shr1"], "dpp/sdwa"
    direct = {"v_cndmask_b32": "cndmask_sgpr", "v_readlane_b32": "readlane", "v_readfirstlane_b32": "readfirstlane",
              "v_writelane_b32": "writelane", "v_permlane32_swap_b32": "permlane32_swap", "v_permlane16_swap_b32": "permlane16_swap",
              "v_subrev_u32": "sub_u32", "v_subb_co_u32": "addc_co_u32", "v_subbrev_co_u32": "addc_co_u32", "v_sub_co_u32": "add_co_u32",
              "v_subrev_co_u32": "add_co_u32", "v_mov_b64": "mov_b64", "v_max_f64": "max_f64", "v_min_u32": "max_u32", "v_max_i32": "min_i32",
              "v_min_f32": "min_f32", "v_max_f32": "max_f32", "v_mul_i32_i24": "mul_u32_u24", "v_mad_i32_i24": "mad_u32_u24",
              "v_mbcnt_hi_u32_b32": "mbcnt_lo", "v_mbcnt_lo_u32_b32": "mbcnt_lo", "v_bfi_b32": "and_or_b32", "v_xad_u32": "add3_u32",
              "v_lshl_add_u64": "lshl_add_u64", "v_mad_u64_u32": "mad_u64_u32", "v_mad_i64_i32": "mad_u64_u32", "v_bfrev_b32": "not_b32",
              "v_ffbh_u32": "cvt_f32_u32", "v_ffbl_b32": "cvt_f32_u32", "v_accvgpr_write_b32": "mov_b32", "v_accvgpr_read_b32": "mov_b32",
              "v_add_f64": "add_f64", "v_mul_f64": "mul_f64", "v_fma_f64": "fma_f64", "v_fmac_f64": "fmac_f64"}
    if base in direct:
        return R[direct[base]], "measured (%s)" % direct[base]
    if name in R:
        return R[name], "measured"
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return R["cmp_le_f64_vcc"] if "f64" in base else R["cmp_lt_u32_vcc"], "measured (compare class)"
    if base.startswith("v_mfma"):
        return 64.0, "matrix pipe (not VALU issue)"
    if "f64" in base or "b64" in base or "u64" in base or "i64" in base:
        if any(t in base for t in ("rcp", "rsq", "sqrt")):
            return R["rcp_f64"], "measured (f64 transcendental class)"
        return R["add_f64"], "assumed: 64-bit class"
    if any(t in base for t in ("exp_f32", "log_f32", "rcp", "rsq", "sqrt", "sin", "cos")):
        return R["exp_f32"], "measured (f32 transcendental class)"
    if base.startswith("v_cvt") or base.startswith("v_frexp") or base.startswith("v_ldexp") or base.startswith("v_rndne") or \
            base.startswith("v_floor") or base.startswith("v_trunc") or base.startswith("v_ceil") or base.startswith("v_fract"):
        return R["cvt_f32_u32"], "measured (conversion class)"
    if base.startswith("v_pk_"):
        return R["pk_add_f32"], "measured (packed class)"
    return R["bfe_u32"], "assumed: half rate (every VOP3 / min / max / shift-left form measured 4 cycles)"


def main():
    R, ghz = rates()
    tmp = "/tmp/valu_mix"
    os.makedirs(tmp, exist_ok=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + OBJ], cwd=tmp, capture_output=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + OBJ,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(tmp, "k.co")], capture_output=True)
    co = os.path.join(tmp, "k.co")
    if not os.path.exists(co) or os.path.getsize(co) == 0:   # older bundler: extract by listing side files
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", OBJ], cwd=tmp, capture_output=True)
        cands = glob.glob(os.path.join(os.path.dirname(OBJ), "tsfa_kernels.o.0.hipv4-*")) + glob.glob(os.path.join(tmp, "*hipv4-*"))
        co = cands[0]
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
    for f in glob.glob(os.path.join(os.path.dirname(OBJ), "tsfa_kernels.o.0.*")):
        os.remove(f)
    cur, hist = None, collections.defaultdict(collections.Counter)
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if m and cur:
            hist[cur][m.group(1)] += 1
    want = {"k_entropy_bits": "_Z14k_entropy_bitsIfLi11EE", "k_entropy": "_Z9k_entropyIfLb1EE", "k_ar": "_Z4k_arIfE", "k_sort": "_Z6k_sortIfE",
            "k_basic": "_Z7k_basicIfE", "k_trend": "_Z7k_trendIfE", "k_spectral": "_Z10k_spectralIfE", "k_seq": "_Z5k_seqIfE",
            "k_cwtpeaks": "_Z10k_cwtpeaksIfLi256EE", "k_cwt_gemm": "_Z10k_cwt_gemmIfLi4EE", "k_stream": "_Z8k_streamIfLi16EE",
            "k_ar_degenerate": "_Z15k_ar_degenerateIfE", "k_langevin_dd": "_Z13k_langevin_dd"}
    doc = {"basis": "static VALU histogram of the float32 instantiation of each kernel (llvm-objdump of tsfa_kernels.o) weighted by "
                    "profiles/r04_issue_rates.jsonl (cycles per wave64 instruction per SIMD at 8 wavefronts per SIMD, measured on the "
                    "box); a PROXY for the dynamic mix", "shader_ghz_measured": ghz, "kernels": {}}
    for k, prefix in want.items():
        h = None
        for name, c in hist.items():
            if name.startswith(prefix):
                h = c
                break
        if h is None:
            continue
        valu = {op: c for op, c in h.items() if op.startswith("v_") and not op.startswith("v_mfma")}
        n = sum(valu.values())
        cyc = 0.0
        by_class = collections.Counter()
        for op, c in valu.items():
            cy, how = classify(op, R)
            cyc += cy * c
            by_class["%.2f" % cy] += c
        doc["kernels"][k] = {"valu_static": n, "cycles_per_valu_inst": cyc / max(n, 1), "salu_static": sum(c for o, c in h.items() if o.startswith("s_")),
                             "lds_static": sum(c for o, c in h.items() if o.startswith("ds_")),
                             "by_cycles": dict(sorted(by_class.items())), "top": [[o, c] for o, c in collections.Counter(valu).most_common(8)]}
    json.dump(doc, open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w"), indent=1)
    for k, v in doc["kernels"].items():
        print("%-16s %6d VALU  %.2f cycles/inst   %s" % (k, v["valu_static"], v["cycles_per_valu_inst"], v["by_cycles"]))
    print("measured shader clock %.3f GHz" % ghz)


if __name__ == "__main__":
    main()
