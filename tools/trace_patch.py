"""Debug aid: instrument conv_igemm.hip with s_memtime trace points (workgroup 7, thread 0) -- apply, build, run
tools/trace_conv.py, then `git checkout mfcnet-tracker_amd/csrc/conv_igemm.hip` and rebuild."""
import os, sys
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mfcnet-tracker_amd", "csrc", "conv_igemm.hip")
s = open(p).read()
i = s.index("template <typename T, int NT, int MT, int PMAX>\n__global__")
s = s[:i] + "__device__ long long g_trace[4096];\n" + s[i:]
s = s.replace("    const int tid = threadIdx.x, lane = tid & 63;", "    int tr_n = 0;\n#define TRACE(tag) do { if (blockIdx.x == 7 && threadIdx.x == 0 && tr_n < 2000) { g_trace[tr_n * 2] = (tag); g_trace[tr_n * 2 + 1] = (long long)__builtin_amdgcn_s_memtime(); ++tr_n; } } while (0)\n    TRACE(0);\n    const int tid = threadIdx.x, lane = tid & 63;", 1)
s = s.replace("    int tl = 0, c = 0, a = 0;          // current stage", "    TRACE(1);\n    int tl = 0, c = 0, a = 0;          // current stage")
s = s.replace("        if (nxt && !(p.ablate & 1)) dma_w(", "        TRACE(10);\n        if (nxt && !(p.ablate & 1)) dma_w(")
s = s.replace("        if (newpatch && !(p.ablate & 2)) load_patch(uc2, c2);", "        TRACE(11);\n        if (newpatch && !(p.ablate & 2)) load_patch(uc2, c2);\n        TRACE(12);")
s = s.replace("        if (red_pending) stats_flush();\n\n        // ---------------- compute stage (tl, c, a) ----------------", "        if (red_pending) stats_flush();\n        TRACE(2);\n        // ---------------- compute stage (tl, c, a) ----------------")
s = s.replace("        // ---------------- tile epilogue ----------------", "        TRACE(3);\n        // ---------------- tile epilogue ----------------")
s = s.replace("        // ---------------- hand over to the next stage ----------------\n        if (newpatch) {\n            __syncthreads();           // every wave has finished reading the current patch\n            if (!(p.ablate & 2)) store_patch();\n        }\n        dma_wait();                    // the weight DMA of stage g+1 has landed (this wave's pieces) ...\n        __syncthreads();               // ... and everybody's",
              "        TRACE(4);\n        if (newpatch) {\n            __syncthreads();\n            TRACE(5);\n            if (!(p.ablate & 2)) store_patch();\n            TRACE(6);\n        }\n        dma_wait();\n        TRACE(7);\n        __syncthreads();\n        TRACE(8);")
s = s.replace("    if (red_pending) stats_flush();\n}\n\n// ------------------------------------------------------------------------------------------\nstatic void choose_tile", "    if (red_pending) stats_flush();\n    TRACE(9);\n    if (blockIdx.x == 7 && threadIdx.x == 0) g_trace[4095] = tr_n;\n}\n\nextern \"C\" int mfc_conv_trace_read(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(long long) * 4096) == hipSuccess ? 0 : -1; }\n\n// ------------------------------------------------------------------------------------------\nstatic void choose_tile")
s = s.replace("                const bool full = (i0 + p.TH <= p.Hl) && (j0 + p.TW <= p.Wl);", "                TRACE(13);\n                const bool full = (i0 + p.TH <= p.Hl) && (j0 + p.TW <= p.Wl);")
s = s.replace("                    const bool vpx = pin[mt] && (full ||", "                    TRACE(14);\n                    const bool vpx = pin[mt] && (full ||")
assert s.count("TRACE(") >= 13, s.count("TRACE(")
open(p, "w").write(s)
print("instrumented", p)
