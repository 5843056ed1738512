/* Plain-C restatement of the MoE indexing + expert FFN contract.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): loaded through ctypes by tests/ as a second, independent checker
 * of the HIP kernels; never linked into the product library.
 *
 * Reference followed (paths relative to /root/reference/TRTAPI++/plugin/fmoe_expert_plugin):
 *   m3o_moe_index      ScatterMappingKernel           fmoe_expert_kernel.cu:25-90
 *   m3o_local_scatter  ScatterMappingCopyKernel       fmoe_expert_kernel.cu:92-118
 *   m3o_local_gather   GatherrMappingCopyKernel       fmoe_expert_kernel.cu:191-217
 *   m3o_expert_ffn     compute_fmoe_expert loop       fmoe_expert_plugin.cpp:82-128
 *                      + BiasSiluKernel / BiasKernel  fmoe_expert_kernel.cu:130-189
 * Rank inside an expert is pinned to the stable order (the reference's atomics leave it
 * unspecified); gate < 0 = dropped row (mapping -1).
 *
 * Build:  gcc -O2 -shared -fPIC -o oracle/_build/libm3oracle.so oracle/moe_index.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

int m3o_moe_index(const int32_t* gate, int S, int E, int32_t* mapping, int32_t* acc) {
  for (int e = 0; e <= E; ++e) acc[e] = 0;
  for (int i = 0; i < S; ++i) {
    int g = gate[i];
    if (g >= E) return -1;
    if (g >= 0) {
      mapping[i] = acc[g + 1]; /* stable rank = arrivals so far */
      acc[g + 1] += 1;
    } else {
      mapping[i] = -1;
    }
  }
  for (int e = 0; e < E; ++e) acc[e + 1] += acc[e];
  for (int i = 0; i < S; ++i)
    if (gate[i] >= 0) mapping[i] += acc[gate[i]];
  return 0;
}

void m3o_local_scatter(const float* x, const int32_t* mapping, int S, int D, float* out) {
  for (int s = 0; s < S; ++s)
    if (mapping[s] >= 0) memcpy(out + (size_t)mapping[s] * D, x + (size_t)s * D, sizeof(float) * D);
}

void m3o_local_gather(const float* buf, const int32_t* mapping, int S, int D, float* out) {
  for (int s = 0; s < S; ++s) {
    if (mapping[s] >= 0)
      memcpy(out + (size_t)s * D, buf + (size_t)mapping[s] * D, sizeof(float) * D);
    else
      memset(out + (size_t)s * D, 0, sizeof(float) * D);
  }
}

/* buf: rows already in scattered order [acc[E], D]; out: same order.  hid: scratch [F]. */
void m3o_expert_ffn(const float* buf, const int32_t* acc, int E, int D, int F, const float* w1,
                    const float* b1, const float* w2, const float* b2, float* hid, float* out) {
  for (int e = 0; e < E; ++e) {
    const float* W1 = w1 + (size_t)e * F * D;
    const float* W2 = w2 + (size_t)e * D * F;
    for (int r = acc[e]; r < acc[e + 1]; ++r) {
      const float* x = buf + (size_t)r * D;
      for (int f = 0; f < F; ++f) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += x[d] * W1[(size_t)f * D + d];
        s += b1[(size_t)e * F + f];
        hid[f] = s / (1.f + expf(-s)); /* SiLU */
      }
      for (int d = 0; d < D; ++d) {
        float s = 0.f;
        for (int f = 0; f < F; ++f) s += hid[f] * W2[(size_t)d * F + f];
        out[(size_t)r * D + d] = s + b2[(size_t)e * D + d];
      }
    }
  }
}
