#!/usr/bin/env python
"""Exposure probe: a build of the library in which named launches can be SKIPPED from the 26th training step on
(SEGMM_SKIP=name[,name...]; results wrong, buffers keep the previous step's valid contents) -- the step time without a kernel
family is an upper bound on what any faster version of it can give the step -- or DUPLICATED (SEGMM_DUP=name[,...], round 6: every
launch of the family is enqueued twice, back to back; the producers of plane tensors are idempotent, so headers, scales and planes stay
valid and the step grows by the family's EXPOSED time -- the number the skip mode cannot give for LayerNorm / l1norm / column sums,
whose consumers fall into their fp32 fallback when the producer is skipped).  The product source is not touched: this script
patches a COPY of csrc/capi.hip into build/probe/ and compiles it to build/probe/libsegmm_skip.so (load it with SEGMM_LIB).
    python tools/probe/build_skip_probe.py && bash tools/probe/skip_table.sh"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "segmminterest_amd", "csrc", "capi.hip")).read()

PRE = r'''
#include <cstdlib>
#include <cstring>
static int g_skip_steps = 0;
static bool skip_on(const char* name) {
    static const char* e = getenv("SEGMM_SKIP");
    if (!e || g_skip_steps < 26) return false;
    const size_t n = strlen(name);
    for (const char* p = e; (p = strstr(p, name)) != nullptr; p += n)
        if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}
#define SEGMM_SKIP(name) do { if (skip_on(name)) return 0; } while (0)
static bool dup_on(const char* name) {
    static const char* e = getenv("SEGMM_DUP");
    if (!e || g_skip_steps < 26) return false;
    const size_t n = strlen(name);
    for (const char* p = e; (p = strstr(p, name)) != nullptr; p += n)
        if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
    return false;
}
'''
ENTRY = {"segmm_attn_fwd": "attn_fwd", "segmm_attn_bwd": "attn_bwd", "segmm_layernorm_fwd": "layernorm_fwd", "segmm_layernorm_fwd_dot": "layernorm_fwd",
         "segmm_layernorm_bwd": "layernorm_bwd", "segmm_layernorm_bwd_outer": "layernorm_bwd", "segmm_layernorm_bwd_pos": "layernorm_bwd",
         "segmm_l1norm": "l1norm", "segmm_colsum": "colsum", "segmm_colsum3": "colsum", "segmm_colsum_pos": "colsum", "segmm_adamw": "adamw",
         "segmm_wsplit_p32": "wsplit", "segmm_label_stats": "loss", "segmm_loss_fwd_bwd": "loss", "segmm_loss_finish": "loss",
         "segmm_site_fixup": "fixup", "segmm_scales_update": "scales"}
out = src
for fn, name in ENTRY.items():
    m = re.search(r"^int %s\(" % fn, out, re.M)
    assert m, fn
    k = out.index(") {\n", m.start())
    # parameter names of the definition (the last identifier of each comma-separated declarator): the duplicate call forwards them
    params = out[out.index("(", m.start()) + 1:k]
    names = [re.findall(r"[A-Za-z_]\w*", d)[-1] for d in params.replace("\n", " ").split(",") if d.strip() and d.strip() != "void"]
    dup = ('    { static thread_local int dup_depth_ = 0; if (dup_depth_ == 0 && dup_on("%s")) { dup_depth_ = 1; const int rc_ = %s(%s); dup_depth_ = 0; '
           'if (rc_) return rc_; } }\n' % (name, fn, ", ".join(names)))
    out = out[:k + 4] + '    SEGMM_SKIP("%s");\n' % name + dup + out[k + 4:]
# the plane GEMM: NT / TN separately, and the split-K combine alone
a = "    if (splits < 1) splits = 1;\n    if (layout == 0) {\n"
g0 = re.search(r"^int segmm_gemm_p\(", out, re.M).start()
k = out.index(a, g0)
out = out[:k] + '    SEGMM_SKIP(layout == 0 ? "gemm_nt" : "gemm_tn");\n' + out[k:]
a = "        hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, s, (const float*)workspace"
k = out.index(a, re.search(r"^int segmm_gemm_p\(", out, re.M).start())
e_ = out.index(";\n", out.index("colsum_out);", k)) + 2 if "colsum_out);" in out[k:k + 600] else out.index(";\n", k) + 2
stmt = out[k:e_]
out = out[:k] + '        if (!skip_on("splitk_reduce")) {\n' + stmt + '        if (dup_on("splitk_reduce")) {\n    ' + stmt + '        }\n        }\n' + out[e_:]
# the step counter
m = re.search(r"^int segmm_step_advance\(", out, re.M)
k = out.index(") {\n", m.start())
out = out[:k + 4] + "    ++g_skip_steps;\n" + out[k + 4:]
# the macro goes behind the last #include of the file
inc = [mm.end() for mm in re.finditer(r'^#include [^\n]*\n', out, re.M) if mm.start() < 4000]
out = out[:inc[-1]] + PRE + out[inc[-1]:]
bd = os.path.join(ROOT, "build", "probe")
os.makedirs(bd, exist_ok=True)
dst = os.path.join(bd, "capi_skip.hip")
open(dst, "w").write(out)
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "segmminterest_amd", "csrc"), "-o", os.path.join(bd, "libsegmm_skip.so"), dst]
print(" ".join(cmd), flush=True)
sys.exit(subprocess.call(cmd))
