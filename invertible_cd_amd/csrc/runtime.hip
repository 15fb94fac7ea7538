// runtime.hip - native executor of one UNet2DConditionModel evaluation (SD1.5 / SDXL topologies) on gfx950.
//
// This is the body of the reference's `model.unet(...)` / `pipe.unet(...)` call (utils/generation.py:241-244,
// utils/generation_sdxl.py:445-453) rebuilt as: a host-side walk of the architecture that bump-allocates activations
// from a caller-owned arena and enqueues the HIP kernels of this library on ONE stream - no per-op Python, no host
// synchronisation (the reference syncs at t.item() every step).  The p2p plugin (utils/p2p.py:291-386) is served by a C
// callback invoked in module-execution order between the probability kernel and the P.V kernel.
//
// Layout: activations fp16 token-major [B, H*W, C]; skip concat is folded into the consumers' loaders (never
// materialised); V is produced transposed by the to_v GEMM epilogue; all ResnetBlock2D time projections are ONE GEMM.
#include <string>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <string.h>
#include "common.h"

namespace {

struct Tensor { const void* ptr; int dtype; long long numel; };

// ---- per-family launch timing (HIP events on the launch stream) ------------------------------------------------
struct ProfRec { hipEvent_t a, b; int kind; double flops, bytes; int M, N, K, aux; int tile_m, tile_n, plan_flags, ksplit; double xflops; };
bool g_prof_on = false;
unsigned g_prof_mask = 0xffffffffu;     // bit k: record launches of family k
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_ev_pool;

hipEvent_t prof_event() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    bool on; hipStream_t st; size_t idx;
    ProfScope(bool active, hipStream_t s, int kind, double flops, double bytes, int M = 0, int N = 0, int K = 0, int aux = 0)
        : on(active && g_prof_on && ((g_prof_mask >> kind) & 1u)), st(s), idx(0) {
        if (!on) return;
        ProfRec r{prof_event(), prof_event(), kind, flops, bytes, M, N, K, aux, 0, 0, 0, 0, flops};
        (void)hipEventRecord(r.a, st);
        idx = g_prof.size();
        g_prof.push_back(r);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(g_prof[idx].b, st); }
    void executed(double xf) { if (on) g_prof[idx].xflops = xf; }      // flops issued to the matrix cores when they differ from the algorithmic count
    void plan(const icd_gemm_desc& d) {          // which tile the planner takes for this launch (tests assert the code path)
        if (!on) return;
        icd_gemm_plan_info pi;
        if (icd_gemm_plan(&d, &pi) != ICD_OK) return;
        ProfRec& r = g_prof[idx];
        r.tile_m = pi.tile_m; r.tile_n = pi.tile_n; r.ksplit = pi.ksplit;
        r.plan_flags = (pi.kernel ? 1 : 0) | (pi.ln_inline ? 2 : 0) | (pi.xattn ? 4 : 0);
    }
};

struct Arena {
    char* base = nullptr;
    long long cap = 0, peak = 0;
    bool dry = false;
    struct Blk { long long off, size; };
    std::vector<Blk> free_list;   // sorted by offset
    std::vector<Blk> live;
    void reset(void* b, long long c, bool d) {
        base = (char*)b; cap = c; dry = d; peak = 0;
        free_list.clear(); live.clear();
        free_list.push_back({0, d ? (1LL << 60) : c});
    }
    void* alloc(long long bytes) {
        bytes = (bytes + 255) & ~255LL;
        if (bytes == 0) bytes = 256;
        for (size_t i = 0; i < free_list.size(); ++i) {
            if (free_list[i].size >= bytes) {
                const long long off = free_list[i].off;
                free_list[i].off += bytes; free_list[i].size -= bytes;
                if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
                live.push_back({off, bytes});
                peak = std::max(peak, off + bytes);
                return (dry ? (char*)0x100000 : base) + off;
            }
        }
        return nullptr;
    }
    void release(void* p) {
        if (!p) return;
        const long long off = (char*)p - (dry ? (char*)0x100000 : base);
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i].off == off) {
                Blk b = live[i];
                live.erase(live.begin() + i);
                auto it = std::lower_bound(free_list.begin(), free_list.end(), b,
                                           [](const Blk& x, const Blk& y) { return x.off < y.off; });
                it = free_list.insert(it, b);
                if (it + 1 != free_list.end() && it->off + it->size == (it + 1)->off) {
                    it->size += (it + 1)->size; free_list.erase(it + 1);
                }
                if (it != free_list.begin() && (it - 1)->off + (it - 1)->size == it->off) {
                    (it - 1)->size += it->size; free_list.erase(it);
                }
                return;
            }
    }
};

}  // namespace

struct icd_unet {
    icd_unet_config cfg;
    std::unordered_map<std::string, Tensor> tensors;
    bool finalized = false;
    int n_attn = 0;
    int temb_total = 0;
    int kv_total = 0;               // sum of C over every transformer block (cross-attention K / V columns)
    // per-handle execution options (icd_unet_set_option)
    // xattn_mode: run LN2 -> to_q -> cross-attention as ONE launch (icd_gemm xattn_*).  0 never, 1 wherever the kernel is eligible,
    // 2 (default) where it measured faster than projection + attention: hosted on the 256-wide tiles (C % 256 == 0), from 4 images per
    // GPU on SDXL's 1024-token layers up to two rounds of 192 x 256 blocks (round 5, profiles/r05_xattn_bench.txt: B = 4 40.5 us against
    // 46.1, B = 8 47.7 against 62.9, B = 16 89.6 against 92.2; the 256 x 256 host alone won at B = 8 only).  On the 128-wide host
    // (C = 640) its softmax epilogue (VALU bound, serial behind the main loop) costs more than the q round trip it saves; DESIGN.md section 4.
    int xattn_mode = 2;
    bool ln_inline = true;          // the GEMM behind a LayerNorm computes its statistics (ICD_GEMM_LN_COMPUTE)
    int xattn_tile = 0;             // A/B: host tile of the fused launch (icd_gemm_desc.tune_xattn_tile)
    bool attn_mode0 = false;        // A/B: flash attention with the scale / offset FMA on the VALU (ICD_ATTN_TUNE_MODE0)
    // Precision of the residual stream.  Every chain x <- x + f(x) of the UNet (ResnetBlock2D: conv2 + input / shortcut;
    // BasicTransformerBlock: the three branch adds; Transformer2DModel: proj_out + input) rounds the whole stream to fp16 once per add,
    // ~100 - 300 adds deep: the dominant error term of an fp16-storage pipeline (eps rel-L2 vs the fp32 oracle 1.1e-3).
    //   0  fp16 stream.
    //   1  fp32 twin (round 3): the chain accumulates in fp32 beside the fp16 copy every consumer reads - 0.7e-3, 6 more bytes per
    //      element and add (-9 % SD1.5, -7 % SDXL).
    //   2  error carry (round 4): one bf8 byte per element holds what the fp16 rounding lost (icd_gemm_desc.resid_carry /
    //      out_carry) - the same 0.7e-3 for 2 more bytes per element and add.
    //   3  carry + split consumers (round 5, default).  With mode 2 the stream itself is accurate, but its CONSUMERS still read the fp16
    //      part alone, and tests/error_budget_sim.py attributes 60 % of the remaining eps variance to exactly that (profiles/
    //      r05_error_budget.txt): GroupNorm inputs 25 + 11 % (stream / skips), the shortcut conv of the channel-changing resnets 20 %,
    //      proj_out 10 %, the sampler convs 8 %; the LayerNorm-folded projections - the expensive readers - only 2 %.  Mode 3:
    //      every GroupNorm normalises fp16 + carry (one more byte per element on its apply pass; conv1 and the sampler convs hand
    //      their outputs on with a carry for it), and the shortcut conv, proj_out and the downsampler conv read hi + lo as a
    //      two-source GEMM over [x | lo] against [W | W] (lo = fp16(2^-14 carry), written by the GroupNorm that reads the same tensor
    //      or by icd_carry_expand): +2 % of the UNet's flops on its cheapest GEMMs.  eps 0.69e-3 -> 0.40e-3 in the simulation.
    int resid_mode = 3;
    // which consumers mode 3 covers (ICD_UNET_OPT_SPLIT_MASK; the error budget of profiles/r05_error_budget.txt toggles them one at a time)
    int split_mask = ICD_SPLIT_DEFAULT;
    // Upsample2D (nearest 2x + conv3x3) as four 2 x 2 convs on the input grid with tap-summed weights (icd_gemm_desc.conv_ktaps):
    // 4/9 of the flops of the 3 x 3 conv on the upsampled map - 8.5 % of an SD1.5 forward's flops become 3.8 %.  0: the 3 x 3 form (A/B).
    bool up_phases = true;
    int gemm_tune = 0;           // ICD_UNET_OPT_GEMM_TUNE: ICD_GEMM_TUNE_* bits OR-ed into every icd_gemm launch of the executor (A/B)
};

namespace {

// token-major activation [B*HW, C]; aux: what the fp16 rounding of the tensor lost - its fp32 twin (residual mode 1) or its bf8 error
// carry (mode 2) - kept only while a consumer will use the tensor as a residual (icd_unet option ICD_UNET_OPT_RESIDUAL_MODE); every
// other consumer (GEMM operands, GroupNorm, skip concat) reads p
struct Act { half_t* p; int C; void* aux = nullptr; };

struct Exec {
    icd_unet* u;
    const icd_unet_io* io;
    hipStream_t st;
    Arena ar;
    bool dry;
    int probs_mode;                          // dry-run materialisation rule (0 none, 1 shipped controllers, 2 all)
    std::vector<std::string>* missing;       // finalize(): collect missing tensor names
    int B, H0, W0, nctx;
    int layer = 0;
    float* gn_ws = nullptr;
    half_t* temb_all = nullptr;
    int temb_off = 0;
    half_t* k_all = nullptr;                 // [B*nctx, kv_total]: K of every cross-attention layer
    half_t* vt_all = nullptr;                // [B, kv_total, ldv_cross]: V^T of every cross-attention layer
    unsigned char* k_all_c = nullptr;        // error carry of k_all (split mode, ICD_SPLIT_QK), or null
    int kv_off = 0;
    bool kv_external = false;
    int status = ICD_OK;
    long long alg_k = 0;                     // algorithmic K of the NEXT gemm_desc launch when it differs from the K it issues (see gemm_desc)

    const void* T(const std::string& name, int dtype, long long numel) {
        auto it = u->tensors.find(name);
        // A forward that has already failed keeps its FIRST error: the walk goes on past a failed launch with stale channel counts, and
        // the tensor it then asks for ("mid_block.resnets.0.conv_shortcut.bias is not bound") used to replace the real message.
        if (!missing && status != ICD_OK) return nullptr;
        if (it == u->tensors.end()) {
            if (missing) { missing->push_back(name); return (const void*)0x1000; }
            icd_set_error("icd_unet_forward: tensor '%s' is not bound", name.c_str());
            status = ICD_ERR_MISSING_TENSOR;
            return nullptr;
        }
        if (it->second.dtype != dtype || it->second.numel != numel) {
            icd_set_error("tensor '%s': expected dtype %d numel %lld, bound dtype %d numel %lld", name.c_str(), dtype, numel,
                          it->second.dtype, it->second.numel);
            status = ICD_ERR_INVALID_ARG;
            return nullptr;
        }
        return it->second.ptr;
    }
    const half_t* Wh(const std::string& n, long long numel) { return (const half_t*)T(n, 0, numel); }
    const float* Wf(const std::string& n, long long numel) { return (const float*)T(n, 1, numel); }

    template <typename TT> TT* alloc(long long elems) {
        void* p = ar.alloc(elems * (long long)sizeof(TT));
        if (!p && status == ICD_OK) {
            icd_set_error("icd_unet_forward: workspace too small (%lld bytes given); query icd_unet_workspace_bytes", ar.cap);
            status = ICD_ERR_WORKSPACE;
        }
        return (TT*)p;
    }
    void release(void* p) { ar.release(p); }
    bool ok() const { return status == ICD_OK; }
    void run(int rc) { if (rc != ICD_OK && status == ICD_OK) status = rc; }

    // ---------------------------------------------------------------------------------------------- op wrappers
    void gemm_desc(icd_gemm_desc& d) {
        // split-K scratch for small-M / deep-K shapes comes from the arena (accounted for in the dry run too)
        void* ws = nullptr;
        const bool splittable = (d.batch <= 1) && !(d.flags & (ICD_GEMM_OUT_TRANS | ICD_GEMM_GEGLU));
        const long long need = splittable ? icd_gemm_workspace_bytes(d.M, d.N, d.K) : 0;
        if (need > 0) { ws = alloc<char>(need); d.splitk_ws = ws; d.splitk_ws_bytes = need; }
        struct Rel { Exec* e; void* p; ~Rel() { if (p) e->release(p); } } rel{this, ws};
        if (!ok() || dry) return;
        const int nb = d.batch > 0 ? d.batch : 1;
        // algorithmic flops = those of the operator as the reference computes it (2 M N K with the 9-tap, unsplit K); the split-operand
        // launches (K doubled by the lo segment) and the phase form of the upsampling conv (4 of 9 taps) issue a different number
        const double xf = 2.0 * d.M * (double)d.N * d.K * nb;
        const double af = alg_k > 0 ? 2.0 * d.M * (double)d.N * alg_k : xf;
        alg_k = 0;
        ProfScope ps(true, st, d.mode == 1 ? ICD_PROF_GEMM_CONV : (nb > 1 ? ICD_PROF_GEMM_BATCHED : ICD_PROF_GEMM_DENSE),
                     af, 0.0, d.M, d.N, d.K, d.mode == 1 ? d.ksize * 100 + d.stride * 10 + d.upsample : d.flags);
        ps.executed(xf);
        d.flags |= u->gemm_tune;
        ps.plan(d);
        run(icd_gemm(&d, st));
    }
    // dense: out[M,N] (ldo) = a[M,K](lda) @ w[N,K]^T + bias + resid
    // resid_aux / out_aux: the fp32 twin or the error carry of `resid` / `out` (residual mode 1 / 2; same leading dimensions)
    void set_aux(icd_gemm_desc& d, const void* resid_aux, void* out_aux) {
        if (u->resid_mode == 1) {
            if (resid_aux) { d.resid = resid_aux; d.flags |= ICD_GEMM_RESID_F32; }       // the fp32 twin IS the residual
            d.out_f32 = (float*)out_aux;
        } else {
            d.resid_carry = resid_aux; d.out_carry = out_aux;
        }
    }
    long long aux_bytes(long long elems) const { return u->resid_mode == 1 ? elems * 4 : elems; }
    void* alloc_aux(long long elems) { return u->resid_mode ? (void*)alloc<char>(aux_bytes(elems)) : nullptr; }
    void linear(const half_t* a, int lda, int M, int K, const half_t* w, int N, const float* bias, const half_t* resid,
                int ldr, half_t* out, int ldo, int flags = 0, int rps = 0, const float* ln_stats = nullptr,
                const float* ln_colsum = nullptr, const void* resid_aux = nullptr, void* out_aux = nullptr) {
        icd_gemm_desc d; memset(&d, 0, sizeof(d));
        d.a0 = a; d.w = w; d.bias = bias; d.resid = resid; d.out = out;
        d.flags = flags;
        set_aux(d, resid_aux, out_aux);
        flags = d.flags;
        d.ln_stats = ln_stats; d.ln_colsum = ln_colsum;
        d.M = M; d.N = N; d.K = K; d.Nw = N; d.lda = lda; d.ldw = K; d.ldo = ldo; d.ldr = ldr;
        d.rows_per_sample = rps; d.mode = 0; d.batch = 1; d.zdiv = 1; d.alpha = 1.f; d.flags = flags;
        gemm_desc(d);
    }
    // resid_aux / out_aux: as in linear(); out_is_f32: `out` itself is float (residual mode 1: the shortcut conv of a ResnetBlock2D,
    // which is only ever a residual)
    // phase: -1, or 2 py + px - one pixel phase of the nearest-2x upsampling conv in its 2 x 2 form on the input grid (icd_gemm_desc.conv_ktaps)
    void conv(const Act& x0, const Act* x1, int Hin, int Win, int ksize, int stride, int upsample, const half_t* w, int Cout,
              const float* bias, const half_t* rowbias, int ld_rowbias, const half_t* resid, void* out, const void* resid_aux = nullptr,
              void* out_aux = nullptr, bool out_is_f32 = false, int phase = -1) {
        icd_gemm_desc d; memset(&d, 0, sizeof(d));
        const int Hu = Hin << upsample, Wu = Win << upsample;
        const int Ho = (Hu + stride - 1) / stride, Wo = (Wu + stride - 1) / stride;
        d.a0 = x0.p; d.a1 = x1 ? x1->p : nullptr; d.w = w; d.bias = bias; d.rowbias = rowbias; d.resid = resid; d.out = out;
        d.C0 = x0.C; d.C1 = x1 ? x1->C : 0;
        d.M = B * Ho * Wo; d.N = Cout; d.K = ksize * ksize * (d.C0 + d.C1); d.Nw = Cout;
        d.ldw = d.K; d.ldo = Cout; d.ldr = Cout; d.ld_rowbias = ld_rowbias; d.rows_per_sample = Ho * Wo;
        d.mode = 1; d.Hin = Hin; d.Win = Win; d.Hout = Ho; d.Wout = Wo; d.ksize = ksize; d.stride = stride; d.upsample = upsample;
        d.batch = 1; d.zdiv = 1; d.alpha = 1.f;
        if (phase >= 0) {
            const int py = phase >> 1, px = phase & 1;
            d.conv_tap_base = 3 * py + px; d.conv_ktaps = 4; d.K = 4 * (d.C0 + d.C1); d.ldw = d.K;
            d.out_remap_w = Wo; d.out_remap_c = 2 * py * Wo + px;
        }
        set_aux(d, resid_aux, out_aux);
        if (out_is_f32) d.flags |= ICD_GEMM_OUT_F32;
        gemm_desc(d);
    }
    void free_act(Act& a) { release(a.p); release(a.aux); a.p = nullptr; a.aux = nullptr; }
    void free_aux(Act& a) { release(a.aux); a.aux = nullptr; }
    // split mode: the inputs' error carries are read by the apply pass; aux (optional) = the [x1 | lo0 | lo1] / [lo0] operand of the
    // resnet's split shortcut conv (icd_groupnorm_carry)
    bool split(int what = ICD_SPLIT_GN) const { return u->resid_mode == 3 && (u->split_mask & what); }
    void groupnorm(const Act& x0, const Act* x1, int HW, const float* g, const float* b, float eps, int silu, half_t* out,
                   half_t* aux = nullptr, int ld_aux = 0) {
        if (!ok() || dry) return;
        const int C = x0.C + (x1 ? x1->C : 0);
        ProfScope ps(true, st, ICD_PROF_GROUPNORM, 0.0, (split() ? 7.0 : 6.0) * B * (double)HW * C + 2.0 * B * (double)HW * (aux ? ld_aux : 0));
        if (split())
            run(icd_groupnorm_carry(x0.p, x0.C, x0.aux, x1 ? x1->p : nullptr, x1 ? x1->C : 0, x1 ? x1->aux : nullptr, B, HW,
                                    u->cfg.norm_groups, g, b, eps, silu, out, aux, ld_aux, gn_ws, st));
        else
            run(icd_groupnorm(x0.p, x0.C, x1 ? x1->p : nullptr, x1 ? x1->C : 0, B, HW, u->cfg.norm_groups, g, b, eps, silu, out, gn_ws, st));
    }
    // lo = fp16(2^-14 carry) of a carried tensor for a split consumer that no GroupNorm precedes
    half_t* expand(const void* carry, long long elems) {
        half_t* lo = alloc<half_t>(elems);
        if (ok() && !dry) {
            ProfScope ps(true, st, ICD_PROF_MISC, 0.0, 3.0 * (double)elems);
            run(icd_carry_expand(carry, elems, lo, st));
        }
        return lo;
    }
    // LayerNorm statistics by a pass over the stream (2 B / element read) - only when the consuming GEMM does not compute them
    void ln_stats(const half_t* x, long long rows, int C, float* stats) {
        if (!ok() || dry || u->ln_inline) return;
        ProfScope ps(true, st, ICD_PROF_LAYERNORM, 0.0, 2.0 * (double)rows * C);
        run(icd_layernorm_stats(x, rows, C, 1e-5f, stats, st));
    }

    // ---------------------------------------------------------------------------------------------- blocks
    Act resnet(const std::string& p, const Act& x0, const Act* x1, int Hh, int Ww, int Cout) {
        const int HW = Hh * Ww, Cin = x0.C + (x1 ? x1->C : 0);
        const long long M = (long long)B * HW;
        half_t* n1 = alloc<half_t>(M * Cin);
        // split mode, channel-changing resnet: norm1 also writes the second source of the split shortcut conv
        const bool split_sc = split(ICD_SPLIT_GN) && split(ICD_SPLIT_SHORTCUT) && Cin != Cout;
        const int ld_sc = x0.C + (x1 ? 2 * x1->C : 0);
        half_t* sc_src = split_sc ? alloc<half_t>(M * ld_sc) : nullptr;
        groupnorm(x0, x1, HW, Wf(p + ".norm1.weight", Cin), Wf(p + ".norm1.bias", Cin), 1e-5f, 1, n1, sc_src, ld_sc);
        half_t* h1 = alloc<half_t>(M * Cout);
        void* h1c = split(ICD_SPLIT_GN) && split(ICD_SPLIT_CONV1) ? alloc_aux(M * Cout) : nullptr;     // conv1's output goes to norm2 with its carry
        Act n1a{n1, Cin};
        conv(n1a, nullptr, Hh, Ww, 3, 1, 0, Wh(p + ".conv1.weight", 9LL * Cin * Cout), Cout, Wf(p + ".conv1.bias", Cout),
             temb_all + temb_off, u->temb_total, nullptr, h1, nullptr, h1c);
        temb_off += Cout;
        release(n1);
        half_t* n2 = alloc<half_t>(M * Cout);
        Act h1a{h1, Cout};
        h1a.aux = h1c;
        groupnorm(h1a, nullptr, HW, Wf(p + ".norm2.weight", Cout), Wf(p + ".norm2.bias", Cout), 1e-5f, 1, n2);
        release(h1); release(h1c);
        const int rm = u->resid_mode;
        const half_t* resid = x0.p;
        const void* resid_aux = rm ? x0.aux : nullptr;
        half_t* sc = nullptr;
        void* sc_aux = nullptr;
        if (Cin != Cout) {
            if (rm == 1) {                           // the shortcut is only ever a residual: fp32 output, no fp16 copy
                sc_aux = alloc<float>(M * Cout);
                conv(x0, x1, Hh, Ww, 1, 1, 0, Wh(p + ".conv_shortcut.weight", (long long)Cin * Cout), Cout,
                     Wf(p + ".conv_shortcut.bias", Cout), nullptr, 0, nullptr, sc_aux, nullptr, nullptr, true);
            } else if (split_sc) {                   // [x0 | x1 | lo0 | lo1] against [W | W]: the shortcut of the values the stream holds
                sc = alloc<half_t>(M * Cout);
                sc_aux = alloc_aux(M * Cout);
                Act s1{sc_src, ld_sc};
                alg_k = Cin;
                conv(x0, &s1, Hh, Ww, 1, 1, 0, Wh(p + ".conv_shortcut.weight2", 2LL * Cin * Cout), Cout,
                     Wf(p + ".conv_shortcut.bias", Cout), nullptr, 0, nullptr, sc, nullptr, sc_aux);
                resid = sc;
            } else {                                 // fp16 (+ its error carry in mode 2)
                sc = alloc<half_t>(M * Cout);
                sc_aux = alloc_aux(M * Cout);
                conv(x0, x1, Hh, Ww, 1, 1, 0, Wh(p + ".conv_shortcut.weight", (long long)Cin * Cout), Cout,
                     Wf(p + ".conv_shortcut.bias", Cout), nullptr, 0, nullptr, sc, nullptr, sc_aux);
                resid = sc;
            }
            release(sc_src);
            resid_aux = sc_aux;
        }
        half_t* out = alloc<half_t>(M * Cout);
        void* out_aux = alloc_aux(M * Cout);
        Act n2a{n2, Cout};
        conv(n2a, nullptr, Hh, Ww, 3, 1, 0, Wh(p + ".conv2.weight", 9LL * Cout * Cout), Cout, Wf(p + ".conv2.bias", Cout),
             nullptr, 0, (rm == 1 && resid_aux) ? nullptr : resid, out, resid_aux, out_aux);
        release(n2);
        release(sc);
        release(sc_aux);
        Act o{out, Cout};
        o.aux = out_aux;
        return o;
    }

    // Phase 0 of the plugin protocol for the next Attention module (module-execution order): does the controller want its
    // probabilities?  Asked BEFORE the query projection is enqueued, because a cross-attention layer that is not
    // materialised runs as the epilogue of that projection (icd_gemm xattn_*).
    struct AttnPlan { int layer; bool mat; void* probs; bool has_epi; icd_probs_epilogue epi; bool half; };
    AttnPlan attn_query(bool is_cross, int place, int heads, int Nq, int Nk) {
        AttnPlan a{layer++, false, nullptr, false, {}, false};
        const long long ldp = (Nk + 7) / 8 * 8;
        if (dry) a.mat = probs_mode == 2 || (probs_mode == 1 && (is_cross || Nq <= 1024));
        else if (io->hook && ok()) {
            void* slots[2] = {nullptr, nullptr};     // [0] the P buffer, [1] an icd_probs_epilogue of this layer call (optional)
            const int r = io->hook(io->hook_user, ICD_HOOK_QUERY, a.layer, is_cross, place, (long long)B * heads, Nq, Nk, ldp, slots);
            if (r < 0) { icd_set_error("attention hook (query) failed at layer %d", a.layer); status = ICD_ERR_HOOK; return a; }
            a.mat = r >= 1;
            a.half = r == 2;                         // P of the second half of the batch only (the conditional rows of a CFG batch)
            if (a.half && (B & 1)) { icd_set_error("attention hook asked for half of an odd batch (layer %d)", a.layer); status = ICD_ERR_HOOK; return a; }
            a.probs = slots[0];
            if (a.mat && slots[1]) { a.epi = *(const icd_probs_epilogue*)slots[1]; a.has_epi = true; }
            if (a.mat && !a.probs) { icd_set_error("attention hook returned 1 without a probability buffer (layer %d)", a.layer); status = ICD_ERR_HOOK; }
        }
        return a;
    }

    // one attention module: q [B*Nq, ldq] (head h at col h*d), k [B*Nk, ldk], vt [B, C, ldv]; out [B*Nq, C]
    // q_c / k_c: error carries of q / k (split mode, materialised layers only), or null
    void attention(const AttnPlan& plan, bool is_cross, int place, const half_t* q, int ldq, const half_t* k, int ldk, const half_t* vt,
                   int ldv, long long vt_bs, int heads, int Nq, int Nk, int d, half_t* out, int C, const void* q_c = nullptr,
                   const void* k_c = nullptr) {
        const int my_layer = plan.layer;
        const long long ldp = (Nk + 7) / 8 * 8;
        // The packer (unet.pack_state_dict) folds d^-1/2 * log2(e) into every query projection: q.k already is the base-2
        // exponent of the softmax.  The flash kernels take it as such (ICD_ATTN_Q_PRESCALED: no scale FMA per score, the running
        // offset subtracted by the MFMA); the kernels with a `scale` argument get ln 2, i.e. scale * log2(e) == 1.
        const float scale = 0.69314718055994531f;
        (void)d;
        const bool mat = plan.mat;
        void* probs = plan.probs;
        if (!ok()) return;
        if (!mat) {
            if (ok() && !dry) {
                ProfScope ps(true, st, ICD_PROF_ATTN_FUSED, 4.0 * B * heads * (double)Nq * Nk * d, 0.0, Nq, Nk, d, heads);
                run(icd_attention_fused_ex(q, k, vt, out, B, heads, Nq, Nk, d, ldq, ldk, ldv, C, vt_bs, scale,
                                           ICD_ATTN_Q_PRESCALED | (u->attn_mode0 ? ICD_ATTN_TUNE_MODE0 : 0), st));
            }
            return;
        }
        // materialised path (utils/p2p.py:335-338): P = softmax(scale q.k^T) as fp16 in ONE pass (icd_attention_probs: no
        // fp32 score tensor) -> hook (the controller edits / keeps P) -> P.V.  plan.half (round 5): the controller works on the second half
        // of the batch only (utils/p2p.py:153-155: attn[h // 2:] of a [uncond; cond] batch) - the first half runs the fused kernel, P is
        // written and read back for the conditional samples alone
        const int b0 = plan.half ? B / 2 : 0, Bm = B - b0;
        if (b0 && !dry) {
            ProfScope ps(true, st, ICD_PROF_ATTN_FUSED, 4.0 * b0 * heads * (double)Nq * Nk * d, 0.0, Nq, Nk, d, heads);
            run(icd_attention_fused_ex(q, k, vt, out, b0, heads, Nq, Nk, d, ldq, ldk, ldv, C, vt_bs, scale,
                                       ICD_ATTN_Q_PRESCALED | (u->attn_mode0 ? ICD_ATTN_TUNE_MODE0 : 0), st));
        }
        const long long qo = (long long)b0 * Nq * ldq, ko = (long long)b0 * Nk * ldk;
        const half_t* qm = q + qo;
        const half_t* km = k + ko;
        const void* qcm = q_c ? (const unsigned char*)q_c + qo : nullptr;
        const void* kcm = k_c ? (const unsigned char*)k_c + ko : nullptr;
        const long long per_b = (long long)heads * Nq * ldp;            // elements of P per sample
        if (!dry && ok()) {
            ProfScope ps(true, st, ICD_PROF_SOFTMAX, 2.0 * Bm * heads * (double)Nq * Nk * d, (double)Bm * per_b * 2.0, Nq, Nk, d,
                         heads | (q_c ? 256 : 0) | (plan.has_epi ? 512 : 0));
            run(icd_attention_probs_ex(qm, qcm, km, kcm, probs, Bm, heads, Nq, Nk, d, ldq, ldk, (int)ldp, scale, plan.has_epi ? &plan.epi : nullptr, st));
        }
        if (!dry && ok()) {
            const int r = io->hook(io->hook_user, ICD_HOOK_PROBS, my_layer, is_cross, place, (long long)Bm * heads, Nq, Nk, ldp, &probs);
            if (r < 0) { icd_set_error("attention hook (probs) failed at layer %d", my_layer); status = ICD_ERR_HOOK; return; }
        }
        icd_gemm_desc g; memset(&g, 0, sizeof(g));
        g.a0 = probs; g.w = vt + (long long)b0 * vt_bs; g.out = out + (long long)b0 * Nq * C;
        g.M = Nq; g.N = d; g.K = (int)ldp; g.Nw = d; g.lda = (int)ldp; g.ldw = ldv; g.ldo = C;
        g.mode = 0; g.batch = Bm * heads; g.zdiv = heads;
        g.a_bs0 = per_b; g.a_bs1 = (long long)Nq * ldp; g.w_bs0 = vt_bs; g.w_bs1 = (long long)d * ldv;
        g.o_bs0 = (long long)Nq * C; g.o_bs1 = d;
        g.alpha = 1.f;
        gemm_desc(g);
    }

    Act transformer(const std::string& p, const Act& x, int Hh, int Ww, int depth, int heads, int place) {
        const int HW = Hh * Ww, C = x.C, d = C / heads, X = u->cfg.cross_dim;
        const long long M = (long long)B * HW;
        const int ldv_self = (HW + 7) / 8 * 8, ldv_cross = (nctx + 7) / 8 * 8;
        half_t* n = alloc<half_t>(M * C);
        groupnorm(x, nullptr, HW, Wf(p + ".norm.weight", C), Wf(p + ".norm.bias", C), 1e-6f, 0, n);
        half_t* h = alloc<half_t>(M * C);
        void* hx = alloc_aux(M * C);                 // fp32 twin / error carry of the transformer blocks' residual stream
        const bool tw = u->resid_mode == 1;          // fp32 twin: the twin IS the residual operand (no fp16 resid beside it)
        linear(n, C, (int)M, C, Wh(p + ".proj_in.weight", (long long)C * C), C, Wf(p + ".proj_in.bias", C), nullptr, 0, h, C, 0, 0,
               nullptr, nullptr, nullptr, hx);
        // the first GEMM behind each LayerNorm computes the statistics of its input rows itself (from its MFMA operand fragments
        // where the tile kernel can, see icd_gemm) and leaves them in lnst for a second consumer (to_v after to_qk)
        const int lnc = u->ln_inline ? ICD_GEMM_LN_COMPUTE : 0;
        release(n);
        // LayerNorm is never materialised: per-row (mean, rstd) from a statistics pass over the residual stream, gamma
        // folded into the consuming projection's weights at load time (unet.py), the rank-1 correction in its epilogue.
        float* lnst = alloc<float>(M * 2);
        for (int kb = 0; kb < depth && ok(); ++kb) {
            const std::string b = p + ".transformer_blocks." + std::to_string(kb);
            // ---- self attention ----
            ln_stats(h, M, C, lnst);
            // (the controller is asked before the projection is enqueued: a layer whose probabilities it keeps gets q and k with their
            //  error carry in split mode - through the general epilogue and a statistics launch, only on such layers)
            const AttnPlan self_plan = attn_query(false, place, heads, HW, HW);
            half_t* qk = alloc<half_t>(M * 2 * C);
            void* qk_c = (self_plan.mat && split(ICD_SPLIT_QK)) ? alloc_aux(M * 2 * C) : nullptr;
            linear(h, C, (int)M, C, Wh(b + ".attn1.to_qk.weight", 2LL * C * C), 2 * C, Wf(b + ".attn1.to_qk.lnbias", 2 * C), nullptr, 0, qk,
                   2 * C, qk_c ? (lnc | ICD_GEMM_TUNE_NO_LN_INLINE) : lnc, 0, lnst, Wf(b + ".attn1.to_qk.lnsum", 2 * C), nullptr, qk_c);
            half_t* vt = alloc<half_t>((long long)B * C * ldv_self);
            linear(h, C, (int)M, C, Wh(b + ".attn1.to_v.weight", (long long)C * C), C, nullptr, nullptr, 0, vt, ldv_self,
                   ICD_GEMM_OUT_TRANS, HW, lnst, Wf(b + ".attn1.to_v.lnsum", C));      // (W_v beta) rides in to_out's bias
            half_t* ao = alloc<half_t>(M * C);
            attention(self_plan, false, place, qk, 2 * C, qk + C, 2 * C, vt, ldv_self, (long long)C * ldv_self, heads, HW, HW, d, ao, C,
                      qk_c, qk_c ? (const unsigned char*)qk_c + C : nullptr);
            release(qk); release(vt); release(qk_c);
            linear(ao, C, (int)M, C, Wh(b + ".attn1.to_out.0.weight", (long long)C * C), C, Wf(b + ".attn1.to_out.0.bias", C), tw ? nullptr : h, C,
                   h, C, 0, 0, nullptr, nullptr, hx, hx);
            // ---- cross attention ----
            ln_stats(h, M, C, lnst);
            // K and V^T of this layer are column / row slices of the per-forward batched projections
            const half_t* kx = k_all + kv_off;
            const half_t* vx = vt_all + (long long)kv_off * ldv_cross;
            const long long vx_bs = (long long)u->kv_total * ldv_cross;
            const AttnPlan cross_plan = attn_query(true, place, heads, HW, nctx);
            const half_t* wq = Wh(b + ".attn2.to_q.weight", (long long)C * C);
            const float* bq = Wf(b + ".attn2.to_q.lnbias", C);
            const float* sq = Wf(b + ".attn2.to_q.lnsum", C);
            const long long xa_blocks = (M / 256) * (C / 256), xa_b192 = (long long)B * ((HW + 191) / 192) * (C / 256);
            const bool xa_auto = C % 256 == 0 && xa_blocks >= 64 && xa_b192 <= 512;
            if (!cross_plan.mat && d == 64 && C % 128 == 0 && HW % 256 == 0 && nctx <= 96 && (u->xattn_mode == 1 || (u->xattn_mode == 2 && xa_auto))) {
                // the north-star kernel: LN2 -> to_q -> softmax(q K^T / 8) V in ONE launch, q stays in the accumulators
                if (ok() && !dry) {
                    icd_gemm_desc g; memset(&g, 0, sizeof(g));
                    g.a0 = h; g.w = wq; g.bias = bq; g.out = ao;
                    g.M = (int)M; g.N = C; g.K = C; g.Nw = C; g.lda = C; g.ldw = C; g.ldo = C;
                    g.rows_per_sample = HW; g.mode = 0; g.batch = 1; g.zdiv = 1; g.alpha = 1.f;
                    g.ln_stats = lnst; g.ln_colsum = sq; g.flags = lnc;
                    g.xattn_k = kx; g.xattn_vt = vx; g.xattn_nk = nctx; g.xattn_ldk = u->kv_total; g.xattn_ldvt = ldv_cross;
                    g.xattn_vt_bs = vx_bs; g.xattn_scale = 0.69314718055994531f; g.tune_xattn_tile = u->xattn_tile;   // q is prescaled (see attention())
                    ProfScope ps(true, st, ICD_PROF_XATTN, 2.0 * M * (double)C * C + 4.0 * M * (double)nctx * C, 0.0, (int)M, C, C, heads);
                    ps.plan(g);
                    run(icd_gemm(&g, st));
                }
            } else {
                half_t* q2 = alloc<half_t>(M * C);
                void* q2_c = (cross_plan.mat && split(ICD_SPLIT_QK)) ? alloc_aux(M * C) : nullptr;
                linear(h, C, (int)M, C, wq, C, bq, nullptr, 0, q2, C, q2_c ? (lnc | ICD_GEMM_TUNE_NO_LN_INLINE) : lnc, 0, lnst, sq, nullptr, q2_c);
                attention(cross_plan, true, place, q2, C, kx, u->kv_total, vx, ldv_cross, vx_bs, heads, HW, nctx, d, ao, C, q2_c,
                          (q2_c && k_all_c) ? k_all_c + kv_off : nullptr);
                release(q2); release(q2_c);
            }
            kv_off += C;
            linear(ao, C, (int)M, C, Wh(b + ".attn2.to_out.0.weight", (long long)C * C), C, Wf(b + ".attn2.to_out.0.bias", C), tw ? nullptr : h, C,
                   h, C, 0, 0, nullptr, nullptr, hx, hx);
            release(ao);
            // ---- GEGLU feed-forward ----
            ln_stats(h, M, C, lnst);
            half_t* ff = alloc<half_t>(M * 4 * C);
            linear(h, C, (int)M, C, Wh(b + ".ff.net.0.proj.weight", 8LL * C * C), 8 * C, Wf(b + ".ff.net.0.proj.bias", 8 * C), nullptr, 0,
                   ff, 4 * C, ICD_GEMM_GEGLU | lnc, 0, lnst, Wf(b + ".ff.net.0.proj.lnsum", 8 * C));
            linear(ff, 4 * C, (int)M, 4 * C, Wh(b + ".ff.net.2.weight", 4LL * C * C), C, Wf(b + ".ff.net.2.bias", C), tw ? nullptr : h, C, h, C,
                   0, 0, nullptr, nullptr, hx, hx);
            release(ff);
        }
        release(lnst);
        half_t* out = alloc<half_t>(M * C);
        void* out_aux = alloc_aux(M * C);
        if (split(ICD_SPLIT_PROJ_OUT)) {             // proj_out over [h | lo] against [W | W] (a two-source 1x1 conv)
            half_t* lo = expand(hx, M * C);
            release(hx);
            Act ha{h, C}, la{lo, C};
            alg_k = C;
            conv(ha, &la, Hh, Ww, 1, 1, 0, Wh(p + ".proj_out.weight2", 2LL * C * C), C, Wf(p + ".proj_out.bias", C), nullptr, 0, x.p, out,
                 x.aux, out_aux);
            release(lo);
        } else {
            release(hx);                             // (proj_out reads the fp16 copy of the stream as its operand)
            linear(h, C, (int)M, C, Wh(p + ".proj_out.weight", (long long)C * C), C, Wf(p + ".proj_out.bias", C), (tw && x.aux) ? nullptr : x.p, C, out, C,
                   0, 0, nullptr, nullptr, x.aux, out_aux);
        }
        release(h);
        Act o{out, C};
        o.aux = out_aux;
        return o;
    }

    // The time-embedding path in fp32 precision (ICD_SPLIT_TEMB): fp32 sinusoids, every Linear over the split operand [hi | lo] of its fp32
    // input against [W | W] with an fp32 output, SiLU on fp32; only temb_all (the per-resnet time biases) is rounded to fp16.  A handful of
    // rows - but an error in emb is the SAME perturbation in all 22 / 17 resnets (tests/error_budget_sim.py: 20 % of what is left on SDXL).
    half_t* split2(const float* x, int rows, int C, int act) {
        half_t* o = alloc<half_t>((long long)rows * 2 * C);
        if (ok() && !dry) run(icd_split2_act(x, rows, C, act, o, st));
        return o;
    }
    int time_embedding_precise() {
        const icd_unet_config& c = u->cfg;
        const int ch0 = c.block_out_channels[0], temb = ch0 * 4;
        const int F32 = ICD_GEMM_OUT_F32, R32 = ICD_GEMM_RESID_F32;
        float* tin = alloc<float>((long long)B * ch0);
        if (ok() && !dry) run(icd_sinusoid_f32(io->timesteps, B, ch0, 0, tin, st));
        if (c.time_cond_proj_dim > 0 && (dry ? true : io->timestep_cond != nullptr))      // + cond_proj(w) (an fp16 input: exact operand)
            linear((const half_t*)io->timestep_cond, c.time_cond_proj_dim, B, c.time_cond_proj_dim,
                   Wh("time_embedding.cond_proj.weight", (long long)ch0 * c.time_cond_proj_dim), ch0, nullptr, (const half_t*)tin, ch0,
                   (half_t*)tin, ch0, F32 | R32);
        half_t* s = split2(tin, B, ch0, 0);
        float* e1 = alloc<float>((long long)B * temb);
        linear(s, 2 * ch0, B, 2 * ch0, Wh("time_embedding.linear_1.weight2", 2LL * temb * ch0), temb, Wf("time_embedding.linear_1.bias", temb),
               nullptr, 0, (half_t*)e1, temb, F32);
        release(s); release(tin);
        s = split2(e1, B, temb, 1);
        float* emb = alloc<float>((long long)B * temb);
        linear(s, 2 * temb, B, 2 * temb, Wh("time_embedding.linear_2.weight2", 2LL * temb * temb), temb, Wf("time_embedding.linear_2.bias", temb),
               nullptr, 0, (half_t*)emb, temb, F32);
        release(s);
        if (c.add_in_dim > 0) {
            const int tdim = c.addition_time_embed_dim, pooled = c.add_in_dim - 6 * tdim;
            if (!dry && (!io->time_ids || !io->text_embeds)) { icd_set_error("SDXL forward needs text_embeds and time_ids"); return ICD_ERR_INVALID_ARG; }
            float* tid = alloc<float>((long long)B * 6 * tdim);
            if (ok() && !dry) run(icd_sinusoid_f32(io->time_ids, B * 6, tdim, 0, tid, st));
            half_t* st2 = split2(tid, B, 6 * tdim, 0);
            release(tid);
            // Linear over cat([text_embeds (fp16 input: exact), hi(time_embeds), lo(time_embeds)]) against [W_text | W_time | W_time]
            Act s0{(half_t*)io->text_embeds, pooled}, s1{st2, 12 * tdim};
            conv(s0, &s1, 1, 1, 1, 1, 0, Wh("add_embedding.linear_1.weight2", (long long)temb * (pooled + 12 * tdim)), temb,
                 Wf("add_embedding.linear_1.bias", temb), nullptr, 0, nullptr, e1, nullptr, nullptr, true);
            release(st2);
            s = split2(e1, B, temb, 1);
            linear(s, 2 * temb, B, 2 * temb, Wh("add_embedding.linear_2.weight2", 2LL * temb * temb), temb, Wf("add_embedding.linear_2.bias", temb),
                   (const half_t*)emb, temb, (half_t*)emb, temb, F32 | R32);
            release(s);
        }
        release(e1);
        s = split2(emb, B, temb, 1);                                  // SiLU(emb), shared by every resnet
        release(emb);
        temb_all = alloc<half_t>((long long)B * u->temb_total);
        linear(s, 2 * temb, B, 2 * temb, Wh("time_emb_proj_cat.weight2", 2LL * u->temb_total * temb), u->temb_total,
               Wf("time_emb_proj_cat.bias", u->temb_total), nullptr, 0, temb_all, u->temb_total);
        release(s);
        return status;
    }

    int forward() {
        const icd_unet_config& c = u->cfg;
        const int L = c.num_levels, ch0 = c.block_out_channels[0], temb = ch0 * 4;
        const int HW0 = H0 * W0;
        gn_ws = alloc<float>(icd_groupnorm_ws_floats(B, HW0, c.norm_groups));
        if (split(ICD_SPLIT_TEMB)) {
            const int rc = time_embedding_precise();
            if (rc != ICD_OK) return rc;
        } else {
        // ---------------- time embedding: Timesteps -> (+cond_proj) -> Linear -> SiLU -> Linear (+ SDXL add_embedding)
        half_t* tsin = alloc<half_t>((long long)B * ch0);
        if (ok() && !dry) run(icd_sinusoid(io->timesteps, B, ch0, 0, tsin, st));
        half_t* tin = tsin;
        if (c.time_cond_proj_dim > 0 && (dry ? true : io->timestep_cond != nullptr)) {
            tin = alloc<half_t>((long long)B * ch0);
            linear((const half_t*)io->timestep_cond, c.time_cond_proj_dim, B, c.time_cond_proj_dim,
                   Wh("time_embedding.cond_proj.weight", (long long)ch0 * c.time_cond_proj_dim), ch0, nullptr, tsin, ch0, tin, ch0);
        }
        half_t* e1 = alloc<half_t>((long long)B * temb);
        linear(tin, ch0, B, ch0, Wh("time_embedding.linear_1.weight", (long long)temb * ch0), temb, Wf("time_embedding.linear_1.bias", temb),
               nullptr, 0, e1, temb);
        if (ok() && !dry) run(icd_silu(e1, (long long)B * temb, e1, st));
        half_t* emb = alloc<half_t>((long long)B * temb);
        linear(e1, temb, B, temb, Wh("time_embedding.linear_2.weight", (long long)temb * temb), temb, Wf("time_embedding.linear_2.bias", temb),
               nullptr, 0, emb, temb);
        if (c.add_in_dim > 0) {
            const int tdim = c.addition_time_embed_dim, pooled = c.add_in_dim - 6 * tdim;
            half_t* tid = alloc<half_t>((long long)B * 6 * tdim);
            if (ok() && !dry) {
                if (!io->time_ids || !io->text_embeds) { icd_set_error("SDXL forward needs text_embeds and time_ids"); return ICD_ERR_INVALID_ARG; }
                run(icd_sinusoid(io->time_ids, B * 6, tdim, 0, tid, st));
            }
            half_t* a1 = alloc<half_t>((long long)B * temb);
            // Linear over cat([text_embeds, time_embeds]) without materialising the concat: 1x1 "conv" with two sources
            Act s0{(half_t*)io->text_embeds, pooled}, s1{tid, 6 * tdim};
            const int saveB = B;
            {
                icd_gemm_desc d; memset(&d, 0, sizeof(d));
                d.a0 = s0.p; d.a1 = s1.p; d.w = Wh("add_embedding.linear_1.weight", (long long)temb * c.add_in_dim);
                d.bias = Wf("add_embedding.linear_1.bias", temb); d.out = a1;
                d.C0 = s0.C; d.C1 = s1.C; d.M = saveB; d.N = temb; d.K = c.add_in_dim; d.Nw = temb; d.ldw = d.K; d.ldo = temb;
                d.rows_per_sample = 1; d.mode = 1; d.Hin = d.Win = d.Hout = d.Wout = 1; d.ksize = 1; d.stride = 1; d.batch = 1; d.zdiv = 1; d.alpha = 1.f;
                gemm_desc(d);
            }
            if (ok() && !dry) run(icd_silu(a1, (long long)B * temb, a1, st));
            linear(a1, temb, B, temb, Wh("add_embedding.linear_2.weight", (long long)temb * temb), temb, Wf("add_embedding.linear_2.bias", temb),
                   emb, temb, emb, temb);
            release(a1); release(tid);
        }
        if (ok() && !dry) run(icd_silu(emb, (long long)B * temb, e1, st));      // e1 := SiLU(emb), shared by every resnet
        temb_all = alloc<half_t>((long long)B * u->temb_total);
        linear(e1, temb, B, temb, Wh("time_emb_proj_cat.weight", (long long)u->temb_total * temb), u->temb_total,
               Wf("time_emb_proj_cat.bias", u->temb_total), nullptr, 0, temb_all, u->temb_total);
        release(e1); release(emb); if (tin != tsin) release(tin); release(tsin);
        }
        temb_off = 0;
        // every cross-attention K / V projection depends on the context only: two GEMMs for the whole forward
        {
            const int X = c.cross_dim, Mc = B * nctx, ldvc = (nctx + 7) / 8 * 8;
            const long long k_elems = (long long)Mc * u->kv_total, v_elems = (long long)B * u->kv_total * ldvc;
            bool have = false;
            if (!dry && io->kv_cache) {                  // caller-owned cache of the context projections (icd_unet_io.kv_cache)
                if (io->kv_cache_bytes < (k_elems + v_elems) * 2 + k_elems) {
                    icd_set_error("icd_unet_forward: kv_cache too small (%lld bytes given); query icd_unet_kv_cache_bytes", (long long)io->kv_cache_bytes);
                    return ICD_ERR_WORKSPACE;
                }
                k_all = (half_t*)io->kv_cache; vt_all = k_all + k_elems; kv_external = true;
                if (split(ICD_SPLIT_QK)) k_all_c = (unsigned char*)(vt_all + v_elems);       // (the cache has room for it: icd_unet_kv_cache_bytes)
                have = io->kv_cache_valid != 0;
            } else {
                k_all = alloc<half_t>(k_elems);
                vt_all = alloc<half_t>(v_elems);
                if (split(ICD_SPLIT_QK)) k_all_c = (unsigned char*)alloc_aux(k_elems);
            }
            if (!have) {
                linear((const half_t*)io->context, X, Mc, X, Wh("attn2_k_cat.weight", (long long)u->kv_total * X), u->kv_total, nullptr,
                       nullptr, 0, k_all, u->kv_total, 0, 0, nullptr, nullptr, nullptr, k_all_c);
                linear((const half_t*)io->context, X, Mc, X, Wh("attn2_v_cat.weight", (long long)u->kv_total * X), u->kv_total, nullptr,
                       nullptr, 0, vt_all, ldvc, ICD_GEMM_OUT_TRANS, nctx);
            } else {                                     // (a finalize-style walk still checks that the weights are bound)
                (void)Wh("attn2_k_cat.weight", (long long)u->kv_total * X); (void)Wh("attn2_v_cat.weight", (long long)u->kv_total * X);
            }
            kv_off = 0;
        }

        // ---------------- conv_in
        std::vector<Act> skips;
        std::vector<int> skipH;
        Act h{alloc<half_t>((long long)B * HW0 * ch0), ch0};
        {
            // conv_in on the matrix cores: pack the 4-channel NCHW latent to [B*HW, 8] and run the implicit GEMM (K = 72)
            half_t* lat8 = alloc<half_t>((long long)B * HW0 * 8);
            if (ok() && !dry) {
                ProfScope ps(true, st, ICD_PROF_MISC, 0.0, (double)B * HW0 * 24.0);
                run(icd_pack_latent(io->sample, io->sample_is_f32, B, HW0, lat8, st));
            }
            Act l8{lat8, 8};
            h.aux = alloc_aux((long long)B * HW0 * ch0);                          // residual of down_blocks.0.resnets.0
            conv(l8, nullptr, H0, W0, 3, 1, 0, Wh("conv_in.weight8", 72LL * ch0), ch0, Wf("conv_in.bias", ch0), nullptr, 0, nullptr, h.p,
                 nullptr, h.aux);
            release(lat8);
        }
        // the skip stack keeps the fp16 tensors (their consumers concatenate them as GEMM / GroupNorm operands); the fp32 twin / error
        // carry of a tensor lives only until the one operator that uses it as a residual has run
        const bool sp = u->resid_mode == 3;
        auto skip_of = [&](const Act& a) { Act s{a.p, a.C}; if (sp) s.aux = a.aux; return s; };   // split mode: skips keep their carry
        skips.push_back(skip_of(h));
        int Hh = H0, Ww = W0;
        // ---------------- down
        for (int i = 0; i < L && ok(); ++i) {
            const int Cout = c.block_out_channels[i];
            for (int j = 0; j < c.layers_per_block && ok(); ++j) {
                const std::string rp = "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
                Act r = resnet(rp, h, nullptr, Hh, Ww, Cout);
                // h stays alive: it is on the skip stack (its twin / carry has served as this resnet's residual)
                if (!sp) free_aux(h);
                h = r;
                if (c.down_has_attn[i]) {
                    Act t = transformer("down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), h, Hh, Ww,
                                        c.transformer_layers[i], c.num_heads[i], 0);
                    free_act(h);
                    h = t;
                }
                skips.push_back(skip_of(h));
            }
            if (i < L - 1) {
                const std::string dp = "down_blocks." + std::to_string(i) + ".downsamplers.0.conv";
                Act dn{alloc<half_t>((long long)B * (Hh / 2) * (Ww / 2) * Cout), Cout};
                // (it is a residual only where the next level keeps the channel count: SD1.5's last level)
                if (split(ICD_SPLIT_SAMPLER_OUT) || c.block_out_channels[i + 1] == Cout) dn.aux = alloc_aux((long long)B * (Hh / 2) * (Ww / 2) * Cout);
                if (split(ICD_SPLIT_DOWN)) {         // the stride-2 conv over [h | lo] against per-tap [W | W]
                    half_t* lo = expand(h.aux, (long long)B * Hh * Ww * Cout);
                    Act la{lo, Cout};
                    alg_k = 9LL * Cout;
                    conv(h, &la, Hh, Ww, 3, 2, 0, Wh(dp + ".weight2", 18LL * Cout * Cout), Cout, Wf(dp + ".bias", Cout), nullptr, 0, nullptr, dn.p,
                         nullptr, dn.aux);
                    release(lo);
                } else {
                    conv(h, nullptr, Hh, Ww, 3, 2, 0, Wh(dp + ".weight", 9LL * Cout * Cout), Cout, Wf(dp + ".bias", Cout), nullptr, 0, nullptr, dn.p,
                         nullptr, dn.aux);
                    if (!sp) free_aux(h);
                }
                Hh /= 2; Ww /= 2;
                h = dn;
                skips.push_back(skip_of(h));
            }
        }
        // ---------------- mid
        {
            const int Cm = c.block_out_channels[L - 1];
            Act r0 = resnet("mid_block.resnets.0", h, nullptr, Hh, Ww, Cm);     // h is the last skip: stays alive
            if (!sp) free_aux(h);
            Act t = transformer("mid_block.attentions.0", r0, Hh, Ww, c.transformer_layers[L - 1], c.num_heads[L - 1], 1);
            free_act(r0);
            Act r1 = resnet("mid_block.resnets.1", t, nullptr, Hh, Ww, Cm);
            free_act(t);
            h = r1;
        }
        // ---------------- up
        for (int i = 0; i < L && ok(); ++i) {
            const int lvl = L - 1 - i;
            const int Cout = c.block_out_channels[lvl];
            for (int j = 0; j < c.layers_per_block + 1 && ok(); ++j) {
                Act sk = skips.back(); skips.pop_back();
                const std::string rp = "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
                Act r = resnet(rp, h, &sk, Hh, Ww, Cout);
                free_act(h); release(sk.p); release(sk.aux);
                h = r;
                if (c.up_has_attn[i]) {
                    Act t = transformer("up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), h, Hh, Ww,
                                        c.transformer_layers[lvl], c.num_heads[lvl], 2);
                    free_act(h);
                    h = t;
                }
                if (!sp) free_aux(h);                // the next resnet concatenates h with a skip: Cin != Cout, its residual is the shortcut
            }                                        // (split mode: its GroupNorm and its split shortcut read the carry)
            if (i < L - 1) {
                const std::string upn = "up_blocks." + std::to_string(i) + ".upsamplers.0.conv";
                Act up{alloc<half_t>((long long)B * (Hh * 2) * (Ww * 2) * Cout), Cout};
                if (split(ICD_SPLIT_SAMPLER_OUT)) up.aux = alloc_aux((long long)B * (Hh * 2) * (Ww * 2) * Cout);
                // (the last upsampler - at the output resolution, damped by nothing downstream - or every one: ICD_SPLIT_UP_ALL)
                const bool sup = split(ICD_SPLIT_UP) && (i == L - 2 || split(ICD_SPLIT_UP_ALL));
                const long long hin = (long long)B * Hh * Ww;
                // (the phase form maps output rows with a divide by the input width: icd_gemm needs out_remap_w >= 2.  A level one pixel wide -
                //  latents of 2^(levels - 1) pixels, which icd_unet_forward accepts - takes the 3 x 3 form, as before round 5)
                const bool phases = u->up_phases && Ww >= 2;
                half_t* lo = nullptr;                // the upsampling conv over [h | lo]
                if (sup && phases) {           // ... in the phase form over [h | lo | h] against [W_hi | W_hi | W_lo] (the tap sums are not fp16 numbers)
                    lo = alloc<half_t>(hin * 2 * Cout);
                    if (ok() && !dry) {
                        ProfScope ps(true, st, ICD_PROF_MISC, 0.0, 7.0 * (double)hin * Cout);
                        run(icd_carry_expand2(h.aux, h.p, hin, Cout, lo, st));
                    }
                } else if (sup) lo = expand(h.aux, hin * Cout);
                Act la{lo, (sup && phases) ? 2 * Cout : Cout};
                if (phases) {
                    // nearest 2x + conv3x3 as four 2 x 2 convs on the input grid, one per output pixel phase: 16 tap GEMMs instead of 36
                    for (int ph = 0; ph < 4 && ok(); ++ph) {
                        const std::string wn = upn + (sup ? ".phase3." : ".phase.") + std::to_string(ph);
                        alg_k = 9LL * Cout;          // per phase: a quarter of the output pixels x all 9 taps of the reference's conv
                        conv(h, sup ? &la : nullptr, Hh, Ww, 3, 1, 0, Wh(wn, (sup ? 12LL : 4LL) * Cout * Cout), Cout, Wf(upn + ".bias", Cout),
                             nullptr, 0, nullptr, up.p, nullptr, up.aux, false, ph);
                    }
                } else {
                    alg_k = 9LL * Cout;
                    conv(h, sup ? &la : nullptr, Hh, Ww, 3, 1, 1, Wh(upn + (sup ? ".weight2" : ".weight"), (sup ? 18LL : 9LL) * Cout * Cout), Cout,
                         Wf(upn + ".bias", Cout), nullptr, 0, nullptr, up.p, nullptr, up.aux);
                }
                release(lo);
                free_act(h);
                Hh *= 2; Ww *= 2;
                h = up;
            }
        }
        // ---------------- out
        half_t* n = alloc<half_t>((long long)B * HW0 * ch0);
        groupnorm(h, nullptr, HW0, Wf("conv_norm_out.weight", ch0), Wf("conv_norm_out.bias", ch0), 1e-5f, 1, n);
        free_act(h);
        const half_t* wo = Wh("conv_out.weight", 9LL * ch0 * c.out_channels);
        const float* bo = Wf("conv_out.bias", c.out_channels);
        if (ok() && !dry) {
            ProfScope ps(true, st, ICD_PROF_MISC, 2.0 * B * (double)HW0 * 9 * ch0 * c.out_channels, 0.0);
            run(icd_conv_out(n, B, H0, W0, ch0, wo, bo, io->eps, io->sample_is_f32, st));
        }
        release(n);
        release(temb_all);
        if (!kv_external) { release(k_all); release(vt_all); release(k_all_c); }
        return status;
    }
};

int count_attn(const icd_unet_config& c) {
    int n = 0;
    const int L = c.num_levels;
    for (int i = 0; i < L; ++i) if (c.down_has_attn[i]) n += c.layers_per_block * c.transformer_layers[i] * 2;
    n += c.transformer_layers[L - 1] * 2;
    for (int i = 0; i < L; ++i) if (c.up_has_attn[i]) n += (c.layers_per_block + 1) * c.transformer_layers[L - 1 - i] * 2;
    return n;
}

int temb_total(const icd_unet_config& c) {
    int t = 0;
    const int L = c.num_levels;
    for (int i = 0; i < L; ++i) t += c.layers_per_block * c.block_out_channels[i];
    t += 2 * c.block_out_channels[L - 1];
    for (int i = 0; i < L; ++i) t += (c.layers_per_block + 1) * c.block_out_channels[L - 1 - i];
    return t;
}

}  // namespace

extern "C" int icd_unet_set_option(icd_unet* u, int32_t option, int32_t value) {
    ICD_CHECK_ARG(u != nullptr, "icd_unet_set_option: null handle");
    switch (option) {
    case ICD_UNET_OPT_XATTN_FUSION:
        ICD_CHECK_ARG(value >= 0 && value <= 2, "icd_unet_set_option: ICD_UNET_OPT_XATTN_FUSION takes 0, 1 or 2 (got %d)", value);
        u->xattn_mode = value; return ICD_OK;
    case ICD_UNET_OPT_XATTN_TILE:
        ICD_CHECK_ARG(value == 0 || value == 2 || value == 4 || value == 5 || value == 6,
                      "icd_unet_set_option: ICD_UNET_OPT_XATTN_TILE takes 0, 2, 4, 5 or 6 (got %d)", value);
        u->xattn_tile = value; return ICD_OK;
    case ICD_UNET_OPT_ATTN_VALU_SCALE:
        ICD_CHECK_ARG(value == 0 || value == 1, "icd_unet_set_option: ICD_UNET_OPT_ATTN_VALU_SCALE takes 0 or 1 (got %d)", value);
        u->attn_mode0 = value != 0; return ICD_OK;
    case ICD_UNET_OPT_RESIDUAL_MODE:
        ICD_CHECK_ARG(value >= 0 && value <= 3, "icd_unet_set_option: ICD_UNET_OPT_RESIDUAL_MODE takes 0 .. 3 (got %d)", value);
        u->resid_mode = value; return ICD_OK;
    case ICD_UNET_OPT_UPSAMPLE_PHASES:
        ICD_CHECK_ARG(value == 0 || value == 1, "icd_unet_set_option: ICD_UNET_OPT_UPSAMPLE_PHASES takes 0 or 1 (got %d)", value);
        u->up_phases = value != 0; return ICD_OK;
    case ICD_UNET_OPT_SPLIT_MASK:
        ICD_CHECK_ARG(value >= 0 && value <= ICD_SPLIT_ALL, "icd_unet_set_option: ICD_UNET_OPT_SPLIT_MASK takes a mask of ICD_SPLIT_* bits (got %d)", value);
        u->split_mask = value; return ICD_OK;
    case ICD_UNET_OPT_LN_INLINE_STATS:
        ICD_CHECK_ARG(value == 0 || value == 1, "icd_unet_set_option: ICD_UNET_OPT_LN_INLINE_STATS takes 0 or 1 (got %d)", value);
        u->ln_inline = value != 0; return ICD_OK;
    case ICD_UNET_OPT_GEMM_TUNE:
        ICD_CHECK_ARG((value & ~(ICD_GEMM_TUNE_NO_PP | ICD_GEMM_TUNE_NO_BIG | ICD_GEMM_TUNE_BN256)) == 0,
                      "icd_unet_set_option: ICD_UNET_OPT_GEMM_TUNE takes ICD_GEMM_TUNE_NO_PP / _NO_BIG / _BN256 bits (got 0x%x)", value);
        u->gemm_tune = value; return ICD_OK;
    }
    icd_set_error("icd_unet_set_option: unknown option %d", option);
    return ICD_ERR_INVALID_ARG;
}

extern "C" int icd_profile_enable(int32_t enable) {
    for (auto& r : g_prof) { g_ev_pool.push_back(r.a); g_ev_pool.push_back(r.b); }
    g_prof.clear();
    g_prof_on = enable != 0;
    g_prof_mask = enable > 0 ? 0xffffffffu : (unsigned)(-enable);     // enable < 0: bit mask of the families to record
    return ICD_OK;
}

extern "C" int icd_profile_read(icd_profile_row* rows, int32_t max_rows) {
    ICD_CHECK_ARG(rows && max_rows >= ICD_PROF_KINDS, "icd_profile_read: need room for %d rows", ICD_PROF_KINDS);
    for (int k = 0; k < ICD_PROF_KINDS; ++k) { rows[k].kind = k; rows[k].launches = 0; rows[k].ms = rows[k].flops = rows[k].bytes = rows[k].flops_executed = 0.0; }
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) { icd_set_error("icd_profile_read: events not complete (synchronise the stream first)"); return ICD_ERR_HIP; }
        rows[r.kind].launches += 1; rows[r.kind].ms += ms; rows[r.kind].flops += r.flops; rows[r.kind].bytes += r.bytes;
        rows[r.kind].flops_executed += r.xflops;
    }
    return ICD_PROF_KINDS;
}

extern "C" int icd_profile_dump(icd_profile_record* recs, int32_t max_recs) {
    const int n = (int)std::min<size_t>(g_prof.size(), (size_t)std::max(0, max_recs));
    for (int i = 0; i < n && recs; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b) != hipSuccess) { icd_set_error("icd_profile_dump: events not complete"); return ICD_ERR_HIP; }
        recs[i].kind = g_prof[i].kind; recs[i].M = g_prof[i].M; recs[i].N = g_prof[i].N; recs[i].K = g_prof[i].K; recs[i].aux = g_prof[i].aux;
        recs[i].ms = ms; recs[i].flops = g_prof[i].flops;
        recs[i].tile_m = g_prof[i].tile_m; recs[i].tile_n = g_prof[i].tile_n; recs[i].plan_flags = g_prof[i].plan_flags; recs[i].ksplit = g_prof[i].ksplit;
    }
    return recs ? n : (int)g_prof.size();
}

extern "C" int icd_unet_create(const icd_unet_config* cfg, icd_unet** out) {
    ICD_CHECK_ARG(cfg && out, "icd_unet_create: null argument");
    ICD_CHECK_ARG(cfg->num_levels >= 2 && cfg->num_levels <= 4, "icd_unet_create: num_levels must be 2..4");
    ICD_CHECK_ARG(cfg->in_channels == 4 && cfg->out_channels == 4, "icd_unet_create: latent channels must be 4");
    ICD_CHECK_ARG(cfg->norm_groups > 0 && cfg->norm_groups <= 64, "icd_unet_create: bad norm_groups");
    for (int i = 0; i < cfg->num_levels; ++i) {
        const int C = cfg->block_out_channels[i];
        ICD_CHECK_ARG(C > 0 && C % 8 == 0 && C % cfg->norm_groups == 0, "icd_unet_create: block_out_channels[%d]=%d unsupported", i, C);
        const int hd = C / std::max(1, cfg->num_heads[i]);
        ICD_CHECK_ARG(cfg->num_heads[i] > 0 && C % cfg->num_heads[i] == 0 && hd % 8 == 0 && hd <= 160,
                      "icd_unet_create: level %d head dim %d unsupported (multiple of 8, <= 160)", i, hd);
    }
    ICD_CHECK_ARG(cfg->cross_dim > 0 && cfg->cross_dim % 8 == 0, "icd_unet_create: cross_dim must be a multiple of 8");
    icd_unet* u = new icd_unet();
    u->cfg = *cfg;
    u->n_attn = count_attn(*cfg);
    u->temb_total = temb_total(*cfg);
    {
        const int L = cfg->num_levels;
        int t = 0;
        for (int i = 0; i < L; ++i) if (cfg->down_has_attn[i]) t += cfg->layers_per_block * cfg->transformer_layers[i] * cfg->block_out_channels[i];
        t += cfg->transformer_layers[L - 1] * cfg->block_out_channels[L - 1];
        for (int i = 0; i < L; ++i) if (cfg->up_has_attn[i]) t += (cfg->layers_per_block + 1) * cfg->transformer_layers[L - 1 - i] * cfg->block_out_channels[L - 1 - i];
        u->kv_total = t;
    }
    *out = u;
    return ICD_OK;
}

extern "C" void icd_unet_destroy(icd_unet* u) { delete u; }

extern "C" int icd_unet_set_tensor(icd_unet* u, const char* name, const void* ptr, int32_t dtype, int64_t numel) {
    ICD_CHECK_ARG(u && name && ptr && numel > 0 && (dtype == 0 || dtype == 1), "icd_unet_set_tensor: bad argument");
    u->tensors[name] = Tensor{ptr, dtype, (long long)numel};
    u->finalized = false;
    return ICD_OK;
}

extern "C" int32_t icd_unet_num_attention_layers(const icd_unet* u) { return u ? u->n_attn : 0; }

static int dry_walk(icd_unet* u, int batch, int H, int W, int nctx, int probs_mode, std::vector<std::string>* missing,
                    long long* peak) {
    Exec e;
    icd_unet_io io; memset(&io, 0, sizeof(io));
    io.batch = batch; io.H = H; io.W = W; io.n_ctx = nctx;
    e.u = u; e.io = &io; e.st = nullptr; e.dry = true; e.probs_mode = probs_mode; e.missing = missing;
    e.B = batch; e.H0 = H; e.W0 = W; e.nctx = nctx;
    e.ar.reset(nullptr, 0, true);
    const int rc = e.forward();
    if (peak) *peak = e.ar.peak;
    return rc;
}

extern "C" int icd_unet_finalize(icd_unet* u) {
    ICD_CHECK_ARG(u, "icd_unet_finalize: null handle");
    std::vector<std::string> missing;
    const int div = 1 << (u->cfg.num_levels - 1);
    const int rc = dry_walk(u, 1, div, div, 8, 0, &missing, nullptr);
    if (!missing.empty()) {
        std::string msg = "icd_unet_finalize: " + std::to_string(missing.size()) + " tensors not bound, first: " + missing[0];
        icd_set_error("%s", msg.c_str());
        return ICD_ERR_MISSING_TENSOR;
    }
    if (rc != ICD_OK) return rc;
    u->finalized = true;
    return ICD_OK;
}

extern "C" int64_t icd_unet_workspace_bytes_ex(const icd_unet* u, int32_t batch, int32_t H, int32_t W, int32_t n_ctx,
                                               int32_t probs_mode) {
    if (!u || batch <= 0 || H <= 0 || W <= 0 || n_ctx <= 0 || probs_mode < 0 || probs_mode > 2) return -1;
    long long peak = 0;
    std::vector<std::string> missing;
    dry_walk(const_cast<icd_unet*>(u), batch, H, W, n_ctx, probs_mode, &missing, &peak);
    return peak + 4096;
}

// Worst case (any hook may ask for the probabilities of any layer).  Since the one-pass probability kernel removed the fp32
// score tensor, the materialisation rule only moves the cross-attention query buffer q2, which never sets the arena's peak:
// the _ex form returns the same number for every probs_mode today and is kept for ABI stability (probabilities themselves
// are the hook's allocations, not arena memory).
extern "C" int64_t icd_unet_workspace_bytes(const icd_unet* u, int32_t batch, int32_t H, int32_t W, int32_t n_ctx) {
    return icd_unet_workspace_bytes_ex(u, batch, H, W, n_ctx, 2);
}

extern "C" int64_t icd_unet_kv_cache_bytes(const icd_unet* u, int32_t batch, int32_t n_ctx) {
    if (!u || batch <= 0 || n_ctx <= 0) return -1;
    const long long ldvc = (n_ctx + 7) / 8 * 8;
    // K [B n_ctx, kv_total] + V^T [B, kv_total, ldvc] in fp16, + one byte per K element for its error carry (ICD_SPLIT_QK)
    return ((long long)batch * n_ctx * u->kv_total + (long long)batch * u->kv_total * ldvc) * 2 + (long long)batch * n_ctx * u->kv_total;
}

extern "C" int icd_unet_forward(icd_unet* u, const icd_unet_io* io, void* stream) {
    ICD_CHECK_ARG(u && io, "icd_unet_forward: null argument");
    ICD_CHECK_ARG(u->finalized, "icd_unet_forward: call icd_unet_finalize first");
    ICD_CHECK_ARG(io->sample && io->timesteps && io->context && io->eps && io->workspace, "icd_unet_forward: null buffer");
    const int div = 1 << (u->cfg.num_levels - 1);
    ICD_CHECK_ARG(io->batch > 0 && io->H > 0 && io->W > 0 && io->H % div == 0 && io->W % div == 0,
                  "icd_unet_forward: H and W must be positive multiples of %d", div);
    ICD_CHECK_ARG(io->n_ctx > 0, "icd_unet_forward: n_ctx must be positive");
    Exec e;
    e.u = u; e.io = io; e.st = (hipStream_t)stream; e.dry = false; e.probs_mode = 0; e.missing = nullptr;
    e.B = io->batch; e.H0 = io->H; e.W0 = io->W; e.nctx = io->n_ctx;
    e.ar.reset(io->workspace, io->workspace_bytes, false);
    return e.forward();
}
