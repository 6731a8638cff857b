#!/usr/bin/env python3
"""Builds a DIAGNOSTIC copy of the HIP library whose k_lsi / k_pip record, per wave and per query
group, when they ran (100 MHz wall clock): build_ab/tl/librayjoin_tl.so.  The product sources are
not touched (the copy lives in the git-ignored build_ab/); the edits are textual and the script
fails loudly when the kernels no longer have the anchors it expects.

  python3 tools/timeline/instrument.py            # copy + patch + make
  gpurun -- python3 tools/timeline/probe.py       # run both kernels, print the distributions

Record layout (uint64 words), one file per kernel launch (RJ_TIMELINE_OUT=<prefix>):
  [0, 8*8192)   per wave: start, first chunk obtained, end, groups done, longest group, XCC id,
                end of the last group, time spent inside next_chunk
  [8*8192, ...) per group (first 2^19 groups): start, (end & 2^40-1) << 24 | wave id << 8 | home part << 4 | XCC id
(k_lsi writes the per-group records only)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "rayjoin_amd", "csrc")
DST = os.path.join(ROOT, "build_ab", "tl")
WAVES, GROUPS = 8192, 1 << 19
WORDS = 8 * WAVES + 2 * GROUPS


def sub(text, old, new, count=1):
    assert text.count(old) >= count, "anchor not found:\n" + old
    return text.replace(old, new, count)


def patch_kernel(k, name, light, group_tail_old, group_tail_new):
    """name: k_lsi / k_pip.  Patches only the text of that kernel.  light: per-group records only
    (k_lsi is capped at 96 SGPRs and 72 VGPRs for its 7 waves per SIMD; per-wave accumulators
    cost it one -- the wave's timeline is rebuilt from its groups, which carry the wave id)."""
    i = k.index("void %s(" % name)
    j = k.index("\n// ====", i) if "\n// ====" in k[i:] else len(k)
    head, body, tail = k[:i], k[i:j], k[j:]
    if not light:
        body = sub(body, "  const long long tk_begin = STATS ? clock64() : 0;\n",
                   "  const long long tk_begin = STATS ? clock64() : 0;\n"
                   "  const unsigned long long tl_t0 = wall_clock64();\n"
                   "  unsigned long long tl_first = 0, tl_n = 0, tl_sched = 0, tl_last = 0;\n")
    # the scheduler call and the start of a group's work (k_lsi: next_chunk + a loop over the chunk;
    # k_pip: next_group hands out one group at a time)
    if name == "k_lsi":
        body = sub(body, "  for (uint64_t g = g_begin; g < g_end; g++) {\n", "  for (uint64_t g = g_begin; g < g_end; g++) {\n@START@")
    else:
        call = "    if (!next_group<kPipWaves>(ranges, wib, A.work_counter, nchunks, A.chunk_groups, ngroups, part, tried, lane, g32)) break;\n"
        body = sub(body, call,
                   "    const unsigned long long tl_s0 = wall_clock64();\n"
                   "    const bool tl_ok = next_group<kPipWaves>(ranges, wib, A.work_counter, nchunks, A.chunk_groups, ngroups, part, tried, lane, g32);\n"
                   "    tl_sched += wall_clock64() - tl_s0;\n"
                   "    if (!tl_ok) break;\n"
                   "    if (!tl_first) tl_first = wall_clock64();\n")
        body = sub(body, "    const uint64_t g = g32;\n", "    const uint64_t g = g32;\n@START@")
    # (no value lives across the group body: the start goes to memory at once)
    body = sub(body, "@START@",
               "    if (!STATS && lane == 0 && g < %dull) {\n"
               "      unsigned long long* tl = *reinterpret_cast<unsigned long long* const*>(A.work_counter + kSchedFaultPtrWord + 2);\n"
               "      if (tl) tl[8 * %d + 2 * g] = wall_clock64();\n"
               "    }\n" % (GROUPS, WAVES))
    acc = "" if light else ("      tl_last = tl_now;\n      tl_n++;\n")
    record = (
        "    {\n"
        "      const unsigned long long tl_now = wall_clock64();\n"
        + acc +
        "      unsigned long long* tl = *reinterpret_cast<unsigned long long* const*>(A.work_counter + kSchedFaultPtrWord + 2);\n"
        "      if (!STATS && tl && lane == 0 && g < %dull) {\n"
        "        unsigned xc;\n"
        "        asm volatile(\"s_getreg_b32 %%0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xc));\n"
        "        tl[8 * %d + 2 * g + 1] = (tl_now << 24) | ((unsigned long long) ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 0xFFFF) << 8) |\n"
        "                                       ((unsigned long long) (blockIdx.x & 7) << 4) | (xc & 0xF);\n"
        "      }\n"
        "    }\n" % (GROUPS, WAVES))
    wave = "" if light else (
        "  unsigned long long* tlw = *reinterpret_cast<unsigned long long* const*>(A.work_counter + kSchedFaultPtrWord + 2);\n"
        "  if (!STATS && tlw && lane == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < %d) {\n"
        "    unsigned long long* o = tlw + 8ull * (blockIdx.x * 4 + (threadIdx.x >> 6));\n"
        "    unsigned xc;\n"
        "    asm volatile(\"s_getreg_b32 %%0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xc));\n"
        "    o[0] = tl_t0; o[1] = tl_first; o[2] = wall_clock64(); o[3] = tl_n; o[4] = 0; o[5] = xc; o[6] = tl_last; o[7] = tl_sched;\n"
        "  }\n" % WAVES)
    body = sub(body, group_tail_old, group_tail_new.replace("@RECORD@", record).replace("@WAVE@", wave))
    if name == "k_lsi":  # the early exit of a group that is clear of the base map
        body = sub(body, "      if (STATS) tk_head += clock64() - tkg;\n      continue;\n",
                   "      if (STATS) tk_head += clock64() - tkg;\n" + record.replace("\n    ", "\n      ").replace("    {", "      {", 1) + "      continue;\n")
    return head + body + tail


def main():
    os.makedirs(DST, exist_ok=True)
    for f in os.listdir(SRC):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            open(os.path.join(DST, f), "w").write(open(os.path.join(SRC, f)).read())
    k = open(os.path.join(DST, "rj_kernels.hip")).read()
    k = patch_kernel(k, "k_lsi", True,
                     "      if (two) process(eb, bb2, pmb);\n    }\n  }\n  }\n",
                     "      if (two) process(eb, bb2, pmb);\n    }\n@RECORD@  }\n  }\n@WAVE@")
    k = patch_kernel(k, "k_pip", False,
                     "    if (STATS) tk_tail += clock64() - tkt;\n  }\n  }\n",
                     "    if (STATS) tk_tail += clock64() - tkt;\n@RECORD@  }\n  }\n@WAVE@")
    # keep k_lsi at its 7 waves per SIMD (the records cost it a register or two: let it spill those)
    k = sub(k, "__global__ __launch_bounds__(256, 6) __attribute__((amdgpu_num_sgpr(96))) void k_lsi(",
            "__global__ __launch_bounds__(256, 7) __attribute__((amdgpu_num_sgpr(96))) void k_lsi(")
    open(os.path.join(DST, "rj_kernels.hip"), "w").write(k)

    a = open(os.path.join(DST, "rj_api.hip")).read()
    hook = (
        "  static unsigned long long* tl_dev = nullptr;\n"
        "  const char* tl_out = getenv(\"RJ_TIMELINE_OUT\");\n"
        "  if (!tl_dev) (void) hipMalloc((void**) &tl_dev, 8ull * %d);\n"
        "  {\n"
        "    static unsigned long long* tl_arg[2];\n"
        "    tl_arg[@SLOT@] = (!h->stats_on && tl_out) ? tl_dev : nullptr;\n"
        "    if (tl_arg[@SLOT@]) (void) hipMemsetAsync(tl_dev, 0, 8ull * %d, @ST@);\n"
        "    (void) hipMemcpyAsync((char*) a.work_counter + (kSchedFaultPtrWord + 2) * 4, &tl_arg[@SLOT@], 8, hipMemcpyHostToDevice, @ST@);\n"
        "  }\n" % (WORDS, WORDS))
    dump = (
        "  if (!h->stats_on && tl_out) {\n"
        "    (void) hipStreamSynchronize(@ST@);\n"
        "    static unsigned long long tl_host[%d];\n"
        "    (void) hipMemcpy(tl_host, tl_dev, sizeof tl_host, hipMemcpyDeviceToHost);\n"
        "    char path[512];\n"
        "    snprintf(path, sizeof path, \"%%s.@KIND@.bin\", tl_out);\n"
        "    FILE* fp = fopen(path, \"wb\");\n"
        "    if (fp) { fwrite(tl_host, 1, sizeof tl_host, fp); fclose(fp); }\n"
        "  }\n" % WORDS)
    a = sub(a, "  tic(h, RJ_T_LSI_KERNEL);  // (after co_pick", hook.replace("@ST@", "h->stream").replace("@SLOT@", "0") + "  tic(h, RJ_T_LSI_KERNEL);  // (after co_pick")
    a = sub(a, "  toc(h, RJ_T_LSI_KERNEL);\n", "  toc(h, RJ_T_LSI_KERNEL);\n" + dump.replace("@ST@", "h->stream").replace("@KIND@", "lsi"))
    a = sub(a, "  tic(h, RJ_T_PIP_KERNEL, st);\n", hook.replace("@ST@", "st").replace("@SLOT@", "1") + "  tic(h, RJ_T_PIP_KERNEL, st);\n")
    a = sub(a, "  toc(h, RJ_T_PIP_KERNEL, st);\n", "  toc(h, RJ_T_PIP_KERNEL, st);\n" + dump.replace("@ST@", "st").replace("@KIND@", "pip"))
    open(os.path.join(DST, "rj_api.hip"), "w").write(a)

    m = open(os.path.join(DST, "Makefile")).read()
    m = sub(m, "OUT     := $(HERE)../librayjoin_amd.so", "OUT     := $(HERE)librayjoin_tl.so")
    open(os.path.join(DST, "Makefile"), "w").write(m)
    subprocess.check_call(["make", "-C", DST, "-j4"])
    print("built", os.path.join(DST, "librayjoin_tl.so"))


if __name__ == "__main__":
    sys.exit(main())
