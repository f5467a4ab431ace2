// A/B timing harness for builds of the tuned forward kernel (measurement helper, not product).
//   fwd_probe [-B 512] [-L 50] [-H 64] [-h 2] [-adv 1] [-rounds 12] [-iters 60] [-stamps N] lib1.so lib2.so ...
// Every lib exports acattn_launch_fwd_dma (a build of ac_tsr_amd/csrc/acattn_fwd_dma.hip, see build_variant.sh).
// Variants are timed in interleaved rounds inside ONE process (HIP events around `iters` back-to-back launches on
// rotating buffer sets larger than the Infinity Cache); outputs of variant k are compared with variant 0.
// -stamps N: the lib was built with -DACATTN_STAMPS; P.noise then carries a [waves][16] u64 stamp buffer, and the
// per-phase distribution over waves (differences of consecutive stamps, in shader cycles) is printed.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "acattn.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef int (*launch_fn)(const acattn_problem&, const acattn_fwd_out&, hipStream_t);
typedef int (*abi_fn)(const acattn_problem*, const acattn_fwd_out*, void*);
typedef int (*sel_fn)(int);
// "libacattn.so:K" = the full library through the C ABI with forward kernel K pinned (acattn_select_forward_kernel)
struct Variant {
  launch_fn direct = nullptr;
  abi_fn abi = nullptr;
  sel_fn sel = nullptr;
  int which = 0;
  bool pre = false;  // "lib.so+pre": the problem carries the producer's affine planes and gate probabilities (ABI 26)
  int operator()(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t st) const {
    if (direct) return direct(p, o, st);
    sel(which);
    return abi(&p, &o, (void*)st);
  }
};

struct Set {
  acattn_problem p;
  acattn_fwd_out o;
  acattn_problem ppre;  // the same problem with affine planes + gate probabilities
};

static float* dev_randn(size_t n, std::mt19937& g, float scale = 1.f, std::vector<float>* keep = nullptr) {
  std::vector<float> h(n);
  std::normal_distribution<float> d(0.f, scale);
  for (auto& x : h) x = d(g);
  float* p;
  CK(hipMalloc(&p, n * 4));
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  if (keep) keep->swap(h);
  return p;
}
static float* dev_copy(const std::vector<float>& h) {
  float* p;
  CK(hipMalloc(&p, h.size() * 4));
  CK(hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  int B = 512, L = 50, H = 64, nh = 2, adv = 1, rounds = 12, iters = 60, nsets = 6, stamps = 0, full_len = 0, balance = 0, causal = 1, where = 0;
  float p_drop = 0.5f, wscale = 0.02f;
  std::vector<std::string> libs;
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* s) { return !strcmp(argv[i], s) && i + 1 < argc; };
    if (is("-B")) B = atoi(argv[++i]);
    else if (is("-L")) L = atoi(argv[++i]);
    else if (is("-H")) H = atoi(argv[++i]);
    else if (is("-h")) nh = atoi(argv[++i]);
    else if (is("-adv")) adv = atoi(argv[++i]);
    else if (is("-rounds")) rounds = atoi(argv[++i]);
    else if (is("-iters")) iters = atoi(argv[++i]);
    else if (is("-sets")) nsets = atoi(argv[++i]);
    else if (is("-stamps")) stamps = atoi(argv[++i]);
    else if (is("-full")) full_len = atoi(argv[++i]);
    else if (is("-balance")) balance = atoi(argv[++i]);
    else if (is("-causal")) causal = atoi(argv[++i]);
    else if (is("-where")) where = atoi(argv[++i]);
    else if (is("-pdrop")) p_drop = atof(argv[++i]);
    else if (is("-wscale")) wscale = atof(argv[++i]);
    else libs.push_back(argv[i]);
  }
  if (libs.empty()) { fprintf(stderr, "no libs\n"); return 2; }
  std::vector<Variant> fns;
  for (auto l : libs) {
    Variant v;
    if (l.size() > 4 && l.substr(l.size() - 4) == "+pre") { v.pre = true; l = l.substr(0, l.size() - 4); }
    std::string path = l;
    const size_t colon = l.rfind(':');
    if (colon != std::string::npos && colon + 2 == l.size()) { path = l.substr(0, colon); v.which = l[colon + 1] - '0'; }
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); return 2; }
    if (path != l) {
      v.abi = (abi_fn)dlsym(h, "acattn_calibrated_attention_fwd");
      v.sel = (sel_fn)dlsym(h, "acattn_select_forward_kernel");
      if (!v.abi || !v.sel) { fprintf(stderr, "no C ABI in %s\n", path.c_str()); return 2; }
    } else {
      void* f = dlsym(h, "_Z21acattn_launch_fwd_dmaRK14acattn_problemRK14acattn_fwd_outP12ihipStream_t");
      if (!f) f = dlsym(h, "_Z24acattn_launch_fwd_streamRK14acattn_problemRK14acattn_fwd_outP12ihipStream_t");
      if (!f) { fprintf(stderr, "no launcher in %s\n", path.c_str()); return 2; }
      v.direct = (launch_fn)f;
    }
    fns.push_back(v);
  }
  std::mt19937 g(42);
  const int dh = H / nh;
  std::vector<float> hwo, hwd, hbo, hbd;
  float* w_order = dev_randn(2 * dh, g, wscale, &hwo);
  float* w_dist = dev_randn(2 * dh, g, wscale, &hwd);
  float* b_order = dev_randn(1, g, wscale, &hbo);
  float* b_dist = dev_randn(1, g, wscale, &hbd);
  float* scalar = dev_randn(1, g, 1.f);
  const size_t n_lh = (size_t)B * L * H, n_ll = (size_t)B * L * L, n_m = (size_t)B * nh * L * L;
  const size_t n_waves = (size_t)B * nh * 4;
  std::vector<Set> sets(nsets);
  unsigned long long* stamp_buf = nullptr;
  if (stamps) { CK(hipMalloc(&stamp_buf, n_waves * 16 * 8)); CK(hipMemset(stamp_buf, 0, n_waves * 16 * 8)); }
  for (int s = 0; s < nsets; ++s) {
    acattn_problem p;
    memset(&p, 0, sizeof p);
    acattn_fwd_out o;
    memset(&o, 0, sizeof o);
    p.B = B; p.L = L; p.H = H; p.n_heads = nh;
    std::vector<float> hq, hk, hg;
    p.q = dev_randn(n_lh, g, 1.f, &hq); p.k = dev_randn(n_lh, g, 1.f, &hk); p.v = dev_randn(n_lh, g);
    std::vector<uint8_t> kv((size_t)B * L);
    std::uniform_int_distribution<int> ld(1, L);
    std::vector<int> lens(B);
    for (int b = 0; b < B; ++b) lens[b] = full_len > 1 ? std::min(full_len, L) : (full_len ? L : ld(g));
    if (balance && B % 128 == 0) {
      // same multiset of lengths, placed so that the 4 sequences that share a CU (b, b+128, ... under the kernel's
      // block decoding and the observed round-robin placement) are a serpentine mix of long and short ones
      std::vector<int> srt = lens;
      std::sort(srt.begin(), srt.end(), std::greater<int>());
      const int Q = B / 4;
      for (int c = 0; c < Q; ++c) {
        lens[c] = srt[c]; lens[c + Q] = srt[2 * Q - 1 - c]; lens[c + 2 * Q] = srt[2 * Q + c]; lens[c + 3 * Q] = srt[4 * Q - 1 - c];
      }
    }
    for (int b = 0; b < B; ++b)
      for (int j = 0; j < L; ++j) kv[(size_t)b * L + j] = j < lens[b];
    uint8_t* kvd;
    CK(hipMalloc(&kvd, kv.size()));
    CK(hipMemcpy(kvd, kv.data(), kv.size(), hipMemcpyHostToDevice));
    p.mask_mode = ACATTN_MASK_STRUCTURED; p.causal = causal; p.key_valid = kvd;
    p.w_order = w_order; p.b_order = b_order; p.w_dist = w_dist; p.b_dist = b_dist; p.scalar = scalar;
    p.adversarial = adv; p.two_level = 1; p.rng_mode = ACATTN_RNG_COUNTER; p.p_drop = p_drop; p.seed = 1234 + s;
    float* cc; CK(hipMalloc(&cc, n_lh * 4)); o.ctx_calibrated = cc;
    if (adv) {
      p.qa = dev_randn(n_lh, g); p.ka = dev_randn(n_lh, g); p.gate_logits = dev_randn(n_ll, g, 1.f, &hg);
      p.combine_option = ACATTN_COMBINE_GATE;
      float *ca, *m, *st;
      CK(hipMalloc(&ca, n_lh * 4)); CK(hipMalloc(&m, n_m * 4)); CK(hipMalloc(&st, (size_t)B * nh * L * ACATTN_NSTAT * 4));
      o.ctx_attacked = ca; o.attack_mask = m; o.row_stats = st;
    }
    if (stamps) p.noise = (const float*)stamp_buf;
    acattn_problem pp = p;
    {  // what the producer of q / k / gate would hand over (acattn.h: affine, gate_is_prob)
      const int LP = 16 * ((L + 15) / 16);
      const double l2e = 1.4426950408889634;
      std::vector<float> aff((size_t)B * nh * 4 * LP, 0.f);
      for (int b = 0; b < B; ++b)
        for (int h = 0; h < nh; ++h)
          for (int i = 0; i < L; ++i) {
            double ao = hbo[0], ad = hbd[0], co = 0, cd = 0;
            for (int d = 0; d < dh; ++d) {
              const double qv = hq[((size_t)b * L + i) * H + h * dh + d], kv2 = hk[((size_t)b * L + i) * H + h * dh + d];
              ao += qv * hwo[d]; ad += qv * hwd[d]; co += kv2 * hwo[dh + d]; cd += kv2 * hwd[dh + d];
            }
            float* pl = &aff[((size_t)b * nh + h) * 4 * LP];
            pl[i] = (float)(-l2e * ao); pl[LP + i] = (float)ad; pl[2 * LP + i] = (float)(-l2e * co); pl[3 * LP + i] = (float)cd;
          }
      pp.affine = dev_copy(aff);
      if (adv) {
        for (auto& x : hg) x = 1.f / (1.f + expf(-x));
        pp.gate_logits = dev_copy(hg);
        pp.gate_is_prob = 1;
      }
    }
    sets[s] = Set{p, o, pp};
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // correctness vs variant 0 on set 0
  std::vector<float> ref_c(n_lh), ref_m(adv ? n_m : 0), cur_c(n_lh), cur_m(adv ? n_m : 0);
  const size_t n_st = adv ? (size_t)B * nh * L * ACATTN_NSTAT : 0;
  std::vector<float> ref_a(adv ? n_lh : 0), cur_a(adv ? n_lh : 0), ref_s(n_st), cur_s(n_st);
  for (size_t v = 0; v < fns.size(); ++v) {
    CK(hipMemsetAsync(sets[0].o.ctx_calibrated, 0xFF, n_lh * 4, st));
    int rc = fns[v](fns[v].pre ? sets[0].ppre : sets[0].p, sets[0].o, st);
    CK(hipStreamSynchronize(st));
    if (rc) { fprintf(stderr, "variant %zu launch rc=%d\n", v, rc); return 3; }
    CK(hipMemcpy(cur_c.data(), sets[0].o.ctx_calibrated, n_lh * 4, hipMemcpyDeviceToHost));
    if (adv) CK(hipMemcpy(cur_m.data(), sets[0].o.attack_mask, n_m * 4, hipMemcpyDeviceToHost));
    if (adv) {
      CK(hipMemcpy(cur_a.data(), sets[0].o.ctx_attacked, n_lh * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(cur_s.data(), sets[0].o.row_stats, n_st * 4, hipMemcpyDeviceToHost));
    }
    if (v == 0) { ref_c = cur_c; ref_m = cur_m; ref_a = cur_a; ref_s = cur_s; }
    double da = 0, ds = 0;
    for (size_t i = 0; i < cur_a.size(); ++i) da = std::max(da, (double)fabsf(cur_a[i] - ref_a[i]));
    for (size_t i = 0; i < cur_s.size(); ++i) ds = std::max(ds, (double)fabsf(cur_s[i] - ref_s[i]));
    printf("variant %zu   max|ctx_att - v0| = %.3e  max|row_stats - v0| = %.3e\n", v, da, ds);
    double dc = 0, dm = 0, sc = 0; size_t nan = 0;
    for (size_t i = 0; i < n_lh; ++i) { if (!(cur_c[i] == cur_c[i])) ++nan; dc = std::max(dc, (double)fabsf(cur_c[i] - ref_c[i])); sc += fabs(cur_c[i]); }
    for (size_t i = 0; i < cur_m.size(); ++i) dm = std::max(dm, (double)fabsf(cur_m[i] - ref_m[i]));
    printf("variant %zu %-40s  max|ctx_cal - v0| = %.3e  max|M - v0| = %.3e  mean|ctx| = %.4f nan=%zu\n", v, libs[v].c_str(), dc, dm, sc / n_lh, nan);
    if (where && v > 0) {  // where the outputs differ from variant 0: (b, head, row, key) of M, (b, row, column) of the contexts
      size_t shown = 0, bad = 0;
      std::vector<size_t> by_qb(16, 0), by_tile(16, 0);
      for (size_t i = 0; i < cur_m.size(); ++i) {
        if (fabsf(cur_m[i] - ref_m[i]) <= 1e-5f) continue;
        const size_t j = i % L, r = (i / L) % L, h = (i / ((size_t)L * L)) % nh, b = i / ((size_t)L * L * nh);
        ++bad; ++by_qb[r >> 4]; ++by_tile[j >> 4];
        if (shown++ < 12) printf("    M[b=%zu h=%zu i=%zu j=%zu] = %.6g  (v0 %.6g)\n", b, h, r, j, cur_m[i], ref_m[i]);
      }
      printf("    M: %zu of %zu elements differ; by query block:", bad, cur_m.size());
      for (int q = 0; q < (L + 15) / 16; ++q) printf(" %zu", by_qb[q]);
      printf("; by key tile:");
      for (int q = 0; q < (L + 15) / 16; ++q) printf(" %zu", by_tile[q]);
      printf("\n");
      for (int which = 0; which < (adv ? 2 : 1); ++which) {
        const std::vector<float>&cu = which ? cur_a : cur_c, &re = which ? ref_a : ref_c;
        size_t nb = 0, sh = 0;
        std::vector<size_t> qb(16, 0);
        std::vector<char> seqbad(B, 0);
        for (size_t i = 0; i < n_lh; ++i) {
          if (fabsf(cu[i] - re[i]) <= 1e-5f && cu[i] == cu[i]) continue;
          const size_t col = i % H, r = (i / H) % L, b = i / ((size_t)H * L);
          ++nb; ++qb[r >> 4]; seqbad[b] = 1;
          if (sh++ < 12) printf("    %s[b=%zu i=%zu col=%zu] = %.6g  (v0 %.6g)\n", which ? "ctx_att" : "ctx_cal", b, r, col, cu[i], re[i]);
        }
        size_t nseq = 0; for (char c2 : seqbad) nseq += c2;
        printf("    %s: %zu elements differ in %zu sequences; by query block:", which ? "ctx_att" : "ctx_cal", nb, nseq);
        for (int q = 0; q < (L + 15) / 16; ++q) printf(" %zu", qb[q]);
        printf("\n");
      }
    }
  }
  std::vector<std::vector<float>> us(fns.size());
  for (int r = -1; r < rounds; ++r) {
    for (size_t v = 0; v < fns.size(); ++v) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) fns[v](fns[v].pre ? sets[i % nsets].ppre : sets[i % nsets].p, sets[i % nsets].o, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 0) us[v].push_back(ms * 1000.f / iters);
    }
  }
  const double bytes = (double)B * ((adv ? 7.0 : 4.0) * 4 * L * H + (adv ? (1.0 + nh) * 4 * L * L : 0.0));
  for (size_t v = 0; v < fns.size(); ++v) {
    std::sort(us[v].begin(), us[v].end());
    const float med = us[v][us[v].size() / 2];
    printf("TIME variant %zu %-40s median %.2f us  min %.2f  max %.2f   -> %.0f GB/s = %.3f of 8 TB/s\n", v, libs[v].c_str(), med,
           us[v].front(), us[v].back(), bytes / med / 1e3, bytes / med / 1e3 / 8000.0);
  }
  if (stamps) {
    // one isolated launch of the LAST variant on set 1, then the stamp statistics
    CK(hipMemset(stamp_buf, 0, n_waves * 16 * 8));
    fns.back()(sets[1 % nsets].p, sets[1 % nsets].o, st);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(n_waves * 16);
    CK(hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t w = 0; w < n_waves; ++w) { if (h[w * 16]) tmin = std::min(tmin, h[w * 16]); for (int k = 0; k < stamps; ++k) tmax = std::max(tmax, h[w * 16 + k]); }
    printf("stamps: launch span %llu cycles (first wave start -> last stamp)\n", tmax - tmin);
    for (int k = 0; k < stamps; ++k) {
      std::vector<double> abs_t, d;
      for (size_t w = 0; w < n_waves; ++w) {
        if (!h[w * 16 + k]) continue;
        abs_t.push_back((double)(h[w * 16 + k] - tmin));
        if (k) d.push_back((double)(h[w * 16 + k] - h[w * 16 + k - 1]));
      }
      if (abs_t.empty()) continue;
      std::sort(abs_t.begin(), abs_t.end());
      std::sort(d.begin(), d.end());
      auto q = [](std::vector<double>& x, double f) { return x.empty() ? 0.0 : x[(size_t)(f * (x.size() - 1))]; };
      printf("  stamp %2d: since launch p10 %7.0f p50 %7.0f p90 %7.0f max %7.0f | since prev p10 %6.0f p50 %6.0f p90 %6.0f max %6.0f (n=%zu)\n", k,
             q(abs_t, .1), q(abs_t, .5), q(abs_t, .9), q(abs_t, 1.0), q(d, .1), q(d, .5), q(d, .9), q(d, 1.0), abs_t.size());
    }
    // streaming kernels: one wave per block, blocks ordered by rank (rank 0 = the heaviest query block under the causal mask)
    for (int rk = 0; rk < 4; ++rk) {
      std::vector<double> tot, st;
      for (size_t w = rk * (n_waves / 4); w < (rk + 1) * (n_waves / 4); ++w)
        if (h[w * 16] && h[w * 16 + stamps - 1]) { tot.push_back((double)(h[w * 16 + stamps - 1] - h[w * 16])); st.push_back((double)(h[w * 16] - tmin)); }
      std::sort(tot.begin(), tot.end()); std::sort(st.begin(), st.end());
      if (!tot.empty()) printf("  rank %d: start p50 %7.0f p90 %7.0f | lifetime p10 %7.0f p50 %7.0f p90 %7.0f\n", rk, st[st.size() / 2], st[(size_t)(0.9 * (st.size() - 1))],
                               tot[(size_t)(0.1 * (tot.size() - 1))], tot[tot.size() / 2], tot[(size_t)(0.9 * (tot.size() - 1))]);
    }
    // by wave index inside the workgroup (query block): p50 of the total
    for (int wv = 0; wv < 4; ++wv) {
      std::vector<double> tot;
      for (size_t w = wv; w < n_waves; w += 4) if (h[w * 16] && h[w * 16 + stamps - 1]) tot.push_back((double)(h[w * 16 + stamps - 1] - h[w * 16]));
      std::sort(tot.begin(), tot.end());
      if (!tot.empty()) printf("  wave %d lifetime p50 %7.0f p90 %7.0f\n", wv, tot[tot.size() / 2], tot[(size_t)(0.9 * (tot.size() - 1))]);
    }
  }
  return 0;
}
