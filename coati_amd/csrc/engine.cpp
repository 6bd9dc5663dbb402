// Engine: the launch sequence of one contrastive + autoregressive training step (reference
// clip_e2e.py:772-814 forward_dist, train_coati.py:237-277 do_minibatch) as host C++ that enqueues the
// gfx950 kernels back to back on one HIP stream.  No device allocation, no host<->device sync.
//
// Memory model (all buffers owned by the caller):
//   params / grads / adam m,v : one flat f32 buffer each; the table below maps the reference's state_dict
//                               names to (offset, rows, cols).  Offsets are 64-element aligned, pads stay zero.
//   shadow                    : bf16 GEMM operands: [0, param_elems) mirrors params elementwise; behind it the
//                               transposed (dgrad) and packed (GNN edge layer) weight copies.
//   workspace                 : saved activations of both transformer passes and the GNN + backward scratch,
//                               carved deterministically per (B, T1, T2, A).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/coati_hip.h"
#include "kernels.h"

static thread_local char g_err[512] = "";
void coati_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* coati_last_error(void) { return g_err; }
extern "C" int coati_abi_version(void) { return COATI_ABI_VERSION; }

namespace {

struct Entry {
  std::string name;
  int64_t off;
  int rows, cols;
};

enum Site {
  SITE_QKV_FWD = 0, SITE_PROJ_FWD, SITE_FC1_FWD, SITE_FC2_FWD, SITE_ATTN_FWD, SITE_LN_FWD, SITE_LMHEAD_FWD,
  SITE_FC2_DGRAD, SITE_FC1_DGRAD, SITE_PROJ_DGRAD, SITE_QKV_DGRAD, SITE_XF_WGRAD, SITE_ATTN_BWD, SITE_LN_BWD,
  SITE_LMHEAD_DLOGITS, SITE_LMHEAD_DGRAD, SITE_LMHEAD_WGRAD, SITE_GNN_EDGE_GEMM, SITE_GNN_NODE_GEMM,
  SITE_GNN_WGRAD, SITE_GNN_ELEMWISE, SITE_EMBED, SITE_OPTIM, SITE_XF_TAIL,
#ifdef COATI_EXPERIMENTAL
  SITE_ATTN_BLOCK_FWD,
#endif
  SITE_COUNT
};
const char* kSiteNames[SITE_COUNT] = {
    "qkv_fwd", "proj_fwd", "fc1_fwd", "fc2_fwd", "attn_fwd", "ln_fwd", "lmhead_fwd", "fc2_dgrad", "fc1_dgrad",
    "proj_dgrad", "qkv_dgrad", "xf_wgrad", "attn_bwd", "ln_bwd", "lmhead_dlogits", "lmhead_dgrad", "lmhead_wgrad",
    "gnn_edge_gemm", "gnn_node_gemm", "gnn_wgrad", "gnn_elemwise", "embed", "optim", "xf_tail",
#ifdef COATI_EXPERIMENTAL
    "attn_block_fwd",
#endif
};   // xf_tail: the [STOP]-row tail of the encoder pass (B-row launches)

struct XLayerP {  // offsets into the flat parameter buffer
  int64_t ln1w, ln1b, attnw, attnb, projw, projb, ln2w, ln2b, fc1w, fc1b, fc2w, fc2b;
  int64_t attnT, projT, fc1T, fc2T;  // offsets into the shadow buffer
  // fp8 mode: byte offsets into the MXFP8 weight buffer: data of [natural | transposed] x {attn, proj, fc1, fc2}, then their scales
  int64_t q8[8], s8[8];
};
struct GLayerP {
  int64_t e0w, e0b, e3w, e3b, n0w, n0b, n3w, n3b, c0w, c0b, c2w;
  int64_t w1ab, w1abT, e3T, n0T, n3T, n0p;  // shadow extras (n0p: residual = True, the [H][2H] part of node_mlp.0.weight packed)
};

struct XPass {  // saved activations of one transformer pass
  int B, T, M;               // M = rows the pass runs on: B * T (padded layout) or the number of packed rows
  int Mcap = 0;              // rows the pass's buffers are carved for (>= B * T: coati_engine_reserve); the cached weight-gradient tables are built on it
  const long long* idx;
  // packed rows (embed.hip launch_seq_pack): the pass runs on the concatenation of every row's real prefix
  bool packed = false;
  int* off = nullptr;        // [B + 1] first packed row of each sequence
  int* ord = nullptr;        // [B] the sequences by descending 16-row block count: launch order of the attention kernels
  int* row_src = nullptr;    // [M] slot b * T + t of packed row m
  int* row_t = nullptr;      // [M] token position of packed row m (rotary embedding)
  long long* ypk = nullptr;  // [M] packed targets (decoder pass)
  int* grp = nullptr;        // [B + 2] work list of the fused attention half (attn_block.hip launch_attn_groups): built by xformer_fwd
  std::vector<float*> x;      // L+1 residual-stream snapshots [M,C]
  std::vector<float*> xmid;   // L
  std::vector<float*> mean1, rstd1, mean2, rstd2, lse;
  std::vector<bf16_t*> a1, qkv, y, a2, g;
  std::vector<unsigned char*> hpre;   // NewGELU'(pre-activation) as 8-bit fixed point (common.h packq8)
  float *meanf, *rstdf, *xf32;
  float *x_emb = nullptr, *mean0 = nullptr, *rstd0 = nullptr;   // norm_embed: the raw embedding rows and their LayerNorm statistics
  bf16_t* af;
  // Encoder pass of a training step: only the [STOP] row of every sequence leaves the transformer (smiles_xformer.py:50-68,
  // clip_e2e.py:448-452), so everything behind the LAST layer's attention + c_proj -- ln_2, the MLP, the residual add and ln_f --
  // runs on those B rows alone ("tail"); their backward likewise.  The rows' buffers:
  bool tail = false;
  float *t_xmid, *t_xL, *t_xf, *t_mean2, *t_rstd2, *t_meanf, *t_rstdf;   // [B, C] f32 / [B]
  bf16_t *t_a2, *t_g;                                                   // [B, C], [B, 4C]
  unsigned char* t_hpre;                                                // [B, 4C]
};

struct Arena {
  char* base;
  size_t off, cap;
  bool dry;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};

}  // namespace

struct coati_engine {
  coati_config cfg;
  std::vector<Entry> entries;
  int64_t n_params = 0, n_shadow = 0, n_trainable = 0;
  int Vpad = 0;
  // parameter offsets
  int64_t tok_emb = 0, lnfw = 0, lnfb = 0, lmhead = 0, lmheadT = 0;
  int64_t emb_lnw = -1, emb_lnb = -1;   // norm_embed: the LayerNorm behind the token embedding (xformer.emb.tok_emb.1.*)
  std::vector<XLayerP> xl;
  std::vector<GLayerP> gl;
  int64_t gembw = 0, gembb = 0, gd0w = 0, gd0b = 0, gd3w = 0, gd3b = 0, gd0T = 0, gd3T = 0;
  int64_t p2c_lnw = 0, p2c_lnb = 0, p2c_w = 0, p2c_b = 0, s2c_lnw = 0, s2c_lnb = 0, s2c_w = 0, s2c_b = 0, tokw = 0, tokb = 0;
  // bound buffers
  float *P = nullptr, *G = nullptr, *Mo = nullptr, *Vo = nullptr;
  bf16_t* S = nullptr;
  unsigned char* S8 = nullptr;     // fp8 mode: MXFP8 copies of the transformer weights (caller-owned, coati_engine_bind_fp8)
  int64_t n_fp8 = 0;               // its size in bytes
  // fp8 mode: two scratch sets (workspace) for quantised operands + scales: [0] = the A operand quantised by gemm8's own pass,
  // [1] = an operand the PREVIOUS product's epilogue emitted (g for FC2, d hidden for the FC1 input gradient)
  unsigned char *q8 = nullptr, *q8s = nullptr, *q8b = nullptr, *q8bs = nullptr;
  const float *cos_t = nullptr, *sin_t = nullptr;
  const int *lut_ix = nullptr, *lut_iy = nullptr;
  // per-step state (pointers into the caller's workspace)
  bool have_fwd = false;
  bool decoder_pending = false;   // a forward stopped behind the heads (train | 2): coati_engine_forward_decoder runs the decoder pass
  bool have_ws = false;    // the workspace is carved (forward or encode): the InfoNCE / optimizer scratch pointers are valid
  int B = 0, T1 = 0, T2 = 0, A = 0;
  XPass p1, p2;
  const long long *y_next = nullptr, *atoms = nullptr;
  const unsigned char* use_point = nullptr;
  float* scal = nullptr;
  // heads
  float *hpoint, *hp_ln, *hp_mean, *hp_rstd, *hstop, *hs_ln, *hs_mean, *hs_rstd, *h_e3gnn, *h_smiles;
  float *sa, *sb, *ptok, *stok, *cliptok, *ones;
  int* stop_pos;
  int* err_flag;
  // lm head
  float2* ce_partial;
  float* ce_lse;
  bf16_t* dlogits;
  // gnn
  std::vector<float*> g_h32, g_rstd;
  std::vector<bf16_t*> g_hcat, g_P, g_e1, g_s2, g_upre, g_t;
  bf16_t *g_hfin16, *g_dpre, *g_td;
  float *g_mask, *g_d2, *g_w, *g_o, *g_o2;
  // compacted edge list (gnn.hip launch_gnn_compact): built once per step on the device, length g_ne[0] never visits the host
  int *g_seg, *g_ne, *g_ebj, *g_ebk, *g_erev, *g_pos;
  float *g_ed2, *g_ew;
  // backward scratch
  float *DX;
  bf16_t *DX16, *g_DO16;
  float *dcliptok, *dptok, *dstok, *dsa, *dsb, *dhe, *dhs, *dhs_ln, *dhstop, *dhp_ln, *dhpoint;
  bf16_t *dh4, *da, *dyb, *dqkv;
  float* attnD;
  float *g_DH, *g_DO;
  bf16_t *g_do2, *g_dtd, *g_du, *g_dmi, *g_ds2, *g_dpre1, *g_dP;
  // deferred transformer weight gradients (grouped launch, gemm.hip wgrad_dma_table_kernel): the bf16 activation gradients
  // of every layer of a pass stay alive until the end of the pass, then ONE launch computes all 4 L weight gradients
  std::vector<bf16_t*> w_dh4, w_dxa, w_dxb, w_dqkv;   // [L]: d(hidden), d x[l+1], d xmid[l], d qkv[l]
  int wg_tile = 128;                                  // output tile of the grouped launch: 256 when C % 256 == 0 (COATI_WGRAD_TILE=128 overrides)
  bool wg_group = false;                              // buffers carved (shape and COATI_WGRAD_GROUP allow it)
  WgradTile* d_wtab = nullptr;                        // device tables: WTAB_SLOTS x (L x tiles per layer) entries
  int wtab_cap = 0;                                   // entries per slot
  struct WTabKey { const void* pass = nullptr; int lo = -1, hi = -1, M = 0, n = 0; long long sig = 0; };
  WTabKey wtab_key[4];
  const void* wtab_ws = nullptr;                      // workspace the cached tables were built for
  long long carve_sig = -1;                           // CAPACITY (B, T1, T2, A) of the last carve: the cached tables die with any other layout
  // grow-only capacities (coati_engine_reserve): every buffer is carved for max(shape of the call, capacity), so that batches whose
  // T / A differ from step to step -- clip_ar_xform truncates every batch to its longest row -- keep the SAME buffer addresses and the
  // cached tables stay valid (rows are a kernel argument of the launches that use them)
  int cap_B = 0, cap_T1 = 0, cap_T2 = 0, cap_A = 0;
  // pinned host staging of the table uploads: 4 transformer slots + 2 for the point encoder, each with the event of its last upload
  // (the uploads are asynchronous: no hipStreamSynchronize on a cache miss)
  WgradTile* h_tab[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t h_tab_cap[6] = {0, 0, 0, 0, 0, 0};
  hipEvent_t h_tab_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int gtab_rr = 0;
  int wtab_rr = 0;                                    // round-robin slot of the next table upload
  // E(3)-GNN node-level weight gradients as ONE split-table launch at the end of gnn_bwd: per-layer copies of the three
  // gradient operands the layers otherwise overwrite, the table (cached like d_wtab)
  // backward of the encoder pass's [STOP]-row tail (XPass::tail)
  float* t_dx = nullptr;                              // [B, C] f32 residual-stream gradient of the tail rows
  bf16_t *t_dxa = nullptr, *t_dxb = nullptr, *t_dh4 = nullptr, *t_da = nullptr;   // [B, C], [B, C], [B, 4C], [B, C]
  bool gnn_wg_group = false;
  std::vector<bf16_t*> gl_DO16, gl_du, gl_dP;
  WgradTile* d_gtab = nullptr;
  int gtab_cap = 0, gtab_n = 0;
  long long gtab_sig = -1;
  const void* gtab_ws = nullptr;
  float* nce = nullptr;
  size_t nce_cap = 0;
  float* opt_partial;
  float* ln_partial;
  float* ln_part_x;   // [2L + 1][COATI_LN_PARTIAL_ROWS][2C]: deferred LayerNorm dgamma / dbeta partials of one transformer pass
  ShadowJob* d_jobs = nullptr;
  int* d_tile_start = nullptr;
  std::vector<ShadowJob> jobs;
  std::vector<int> tile_start;
  int n_job_tiles = 0;
  bool jobs_uploaded = false;
  // side stream: the point encoder (independent of the transformer passes) runs concurrently with them
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool overlap = getenv("COATI_NO_OVERLAP") == nullptr;   // (A/B switch: the point encoder on the main stream)
  bool gnn_bwd_done = false;   // staged backward: stage 2 already ran the point-encoder backward on the side stream
  bool gnn_side_pending = false;   // stage 4 forked it; stage 5 joins
  // profiling
  unsigned long long prof_mask = 0;   // launch sites whose launches are bracketed by HIP events (bit = Site)
  bool prof_paused = false;         // events suspended (the selection, its counters and the overlap setting stay): bench.py samples every 4th step
  bool prof_keep_overlap = false;   // timed-region profiling: events around the selected site, the step itself runs as the product does (side stream on)
  std::vector<hipEvent_t> ev;
  int ev_used = 0;
  double prof_flops = 0.0;
  double prof_last_bytes = 0.0;   // per launch, of the last prof_collect
  // ---- inference decode (KV cache) ----
  struct Decode {
    bool active = false;
    int B = 0, Tmax = 0, pos = 0;
    bf16_t* cache = nullptr;   // [L][B][nh][Tmax][k16|v16]
    float *x = nullptr, *xmid = nullptr, *xn = nullptr, *mean = nullptr, *rstd = nullptr;
    bf16_t *a = nullptr, *qkv = nullptr, *y = nullptr, *hpre = nullptr, *g = nullptr, *af = nullptr;
    // graph replay: position, input tokens / injection and logits live at fixed device addresses
    int* pos_dev = nullptr;
    long long* tok_dev = nullptr;
    float* inj_dev = nullptr;
    float* logits_dev = nullptr;
    int64_t ldl = 0;
    hipGraphExec_t graph[2] = {nullptr, nullptr};   // [0] tokens only, [1] tokens + injection
  } dec;
  double prof_bytes = 0.0;   // algorithmic HBM bytes (operands read once, results written once) of the selected site
  // the E(3)-GNN's edge-level launches: their row count (the compacted neighbour list's length, g_ne[0]) lives on the device, so the
  // sites book bytes and flops PER EDGE; coati_engine_prof_collect reads the count back (it synchronises anyway) and multiplies
  double prof_bytes_e = 0.0, prof_flops_e = 0.0;
};

namespace {

int64_t add_entry(coati_engine* e, const std::string& name, int rows, int cols) {
  const int64_t n = (int64_t)rows * (cols > 0 ? cols : 1);
  const int64_t off = e->n_params;
  e->entries.push_back({name, off, rows, cols});
  e->n_params = (off + n + 63) & ~(int64_t)63;
  return off;
}
int64_t add_shadow(coati_engine* e, int64_t n) {
  const int64_t off = e->n_shadow;
  e->n_shadow = (off + n + 63) & ~(int64_t)63;
  return off;
}

void build_layout(coati_engine* e) {
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer, H = c.n_hidden_e3nn, E = c.n_embd_common, V = c.n_tok;
  e->Vpad = (V + 63) & ~63;
  // --- transformer (smiles_xformer.py:71-100, basic_transformer.py:103-169) ---
  if (c.norm_embed) {   // tok_emb = Sequential(Embedding, LayerNorm) (basic_transformer.py:72-76)
    e->tok_emb = add_entry(e, "xformer.emb.tok_emb.0.weight", V, C);
    e->emb_lnw = add_entry(e, "xformer.emb.tok_emb.1.weight", C, 0);
    e->emb_lnb = add_entry(e, "xformer.emb.tok_emb.1.bias", C, 0);
  } else {
    e->tok_emb = add_entry(e, "xformer.emb.tok_emb.weight", V, C);
  }
  e->xl.resize(c.n_layer_xformer);
  for (int l = 0; l < c.n_layer_xformer; ++l) {
    const std::string p = "xformer.transformer.h." + std::to_string(l) + ".";
    XLayerP& x = e->xl[l];
    x.ln1w = add_entry(e, p + "ln_1.weight", C, 0);
    x.ln1b = add_entry(e, p + "ln_1.bias", C, 0);
    x.attnw = add_entry(e, p + "attn.c_attn.weight", 3 * C, C);
    x.attnb = c.biases ? add_entry(e, p + "attn.c_attn.bias", 3 * C, 0) : -1;
    x.projw = add_entry(e, p + "attn.c_proj.weight", C, C);
    x.projb = c.biases ? add_entry(e, p + "attn.c_proj.bias", C, 0) : -1;
    x.ln2w = add_entry(e, p + "ln_2.weight", C, 0);
    x.ln2b = add_entry(e, p + "ln_2.bias", C, 0);
    x.fc1w = add_entry(e, p + "mlpf.0.weight", 4 * C, C);
    x.fc1b = c.biases ? add_entry(e, p + "mlpf.0.bias", 4 * C, 0) : -1;
    x.fc2w = add_entry(e, p + "mlpf.2.weight", C, 4 * C);
    x.fc2b = c.biases ? add_entry(e, p + "mlpf.2.bias", C, 0) : -1;
  }
  e->lnfw = add_entry(e, "xformer.transformer.ln_f.weight", C, 0);
  e->lnfb = add_entry(e, "xformer.transformer.ln_f.bias", C, 0);
  // Order of the flat buffers (round 4; entries are found by NAME everywhere, tests/test_host_cpu.py): transformer body | point
  // encoder | lm_head | heads.  The data-parallel step then needs TWO gradient collectives instead of four: lm_head + heads are final
  // behind the decoder stage of the backward and adjacent, transformer body + point encoder behind the encoder stage and adjacent
  // (coati_amd/distributed.py grad_buckets; every collective costs a pair of stream hand-overs and a launch on RCCL's stream).
  // --- point encoder (e3gnn_clip.py:75-104, e_gcl_sparse.py:130-150) ---
  auto add_point = [&]() {
    if (c.torch_emb) {   // nn.Embedding(84, H) in front, embedding = nn.Identity (e3gnn_clip.py:49-56, 74-77)
      e->gembw = add_entry(e, "point_encoder.emb.weight", 84, H);
      e->gembb = -1;
    } else {
      e->gembw = add_entry(e, "point_encoder.embedding.weight", H, 28);
      e->gembb = add_entry(e, "point_encoder.embedding.bias", H, 0);
    }
    e->gd0w = add_entry(e, "point_encoder.node_dec.0.weight", H, H);
    e->gd0b = add_entry(e, "point_encoder.node_dec.0.bias", H, 0);
    e->gd3w = add_entry(e, "point_encoder.node_dec.3.weight", H, H);
    e->gd3b = add_entry(e, "point_encoder.node_dec.3.bias", H, 0);
    e->gl.resize(c.n_layer_e3gnn);
    for (int l = 0; l < c.n_layer_e3gnn; ++l) {
      const std::string p = "point_encoder.gcl_" + std::to_string(l) + ".";
      GLayerP& g = e->gl[l];
      g.e0w = add_entry(e, p + "edge_mlp.0.weight", H, 2 * H + 1);
      g.e0b = add_entry(e, p + "edge_mlp.0.bias", H, 0);
      g.e3w = add_entry(e, p + "edge_mlp.3.weight", H, H);
      g.e3b = add_entry(e, p + "edge_mlp.3.bias", H, 0);
      g.n0w = add_entry(e, p + "node_mlp.0.weight", H, 2 * H + (c.residual ? 28 : 0));   // residual: + the one-hot node features (e_gcl_sparse.py:141)
      g.n0b = add_entry(e, p + "node_mlp.0.bias", H, 0);
      g.n3w = add_entry(e, p + "node_mlp.3.weight", H, H);
      g.n3b = add_entry(e, p + "node_mlp.3.bias", H, 0);
    }
  };
  // --- point_to_clip (clip_e2e.py:405-422) ---
  auto add_p2c = [&]() {
    // norm_clips: LayerNorm -> Linear (state_dict .0 / .1); otherwise a plain Linear (clip_e2e.py:405-428)
    if (c.norm_clips && c.old_architecture) {   // Linear -> LayerNorm (clip_e2e.py:409-413)
      e->p2c_w = add_entry(e, "point_to_clip.0.weight", E, H);
      e->p2c_b = add_entry(e, "point_to_clip.0.bias", E, 0);
      e->p2c_lnw = add_entry(e, "point_to_clip.1.weight", H, 0);
      e->p2c_lnb = add_entry(e, "point_to_clip.1.bias", H, 0);
    } else if (c.norm_clips) {
      e->p2c_lnw = add_entry(e, "point_to_clip.0.weight", H, 0);
      e->p2c_lnb = add_entry(e, "point_to_clip.0.bias", H, 0);
      e->p2c_w = add_entry(e, "point_to_clip.1.weight", E, H);
      e->p2c_b = add_entry(e, "point_to_clip.1.bias", E, 0);
    } else {
      e->p2c_w = add_entry(e, "point_to_clip.weight", E, H);
      e->p2c_b = add_entry(e, "point_to_clip.bias", E, 0);
    }
  };
  // use_point_encoder = False (clip_e2e.py:454-463): encode_points returns zeros, so the point encoder and point_to_clip never
  // receive a gradient (p.grad is None: torch's clip_grad_norm_ / AdamW skip them).  Their parameters exist in the state_dict
  // all the same: they go BEHIND n_trainable, next to coord_mlp
  if (c.use_point_encoder) add_point();
  e->lmhead = add_entry(e, "xformer.lm_head.weight", V, C);
  if (c.use_point_encoder) add_p2c();
  // --- heads (clip_e2e.py:419-435) ---
  if (c.norm_clips && c.old_architecture) {   // Linear -> LayerNorm (clip_e2e.py:414-417)
    e->s2c_w = add_entry(e, "smiles_to_clip.0.weight", E, C);
    e->s2c_b = add_entry(e, "smiles_to_clip.0.bias", E, 0);
    e->s2c_lnw = add_entry(e, "smiles_to_clip.1.weight", E, 0);
    e->s2c_lnb = add_entry(e, "smiles_to_clip.1.bias", E, 0);
  } else if (c.norm_clips) {
    e->s2c_lnw = add_entry(e, "smiles_to_clip.0.weight", E, 0);
    e->s2c_lnb = add_entry(e, "smiles_to_clip.0.bias", E, 0);
    e->s2c_w = add_entry(e, "smiles_to_clip.1.weight", E, C);
    e->s2c_b = add_entry(e, "smiles_to_clip.1.bias", E, 0);
  } else {
    e->s2c_w = add_entry(e, "smiles_to_clip.weight", E, C);
    e->s2c_b = add_entry(e, "smiles_to_clip.bias", E, 0);
  }
  if (c.token_mlp) {   // SiLU -> Linear; otherwise nn.Identity (no parameters)
    e->tokw = add_entry(e, "point_clip_to_special_tokens.1.weight", E, E);
    e->tokb = add_entry(e, "point_clip_to_special_tokens.1.bias", E, 0);
  }
  // coord_mlp is evaluated and discarded by the reference (e3gnn_clip.py:132): its parameters never receive a gradient
  // (p.grad is None), so torch's clip_grad_norm_ / AdamW skip them entirely -- no weight decay either.  They are kept for
  // state_dict parity at the END of the flat buffers, behind n_trainable: the optimizer kernels stop in front of them.
  e->n_trainable = e->n_params;
  if (c.norm_embed) {   // the reference registers a second LayerNorm it never calls (smiles_xformer.py:81-82, 364): state_dict parity only
    add_entry(e, "xformer.norm_embed.weight", C, 0);
    add_entry(e, "xformer.norm_embed.bias", C, 0);
  }
  if (!c.use_point_encoder) { add_point(); add_p2c(); }
  for (int l = 0; l < c.n_layer_e3gnn; ++l) {
    const std::string p = "point_encoder.gcl_" + std::to_string(l) + ".";
    GLayerP& g = e->gl[l];
    g.c0w = add_entry(e, p + "coord_mlp.0.weight", H, H);
    g.c0b = add_entry(e, p + "coord_mlp.0.bias", H, 0);
    g.c2w = add_entry(e, p + "coord_mlp.2.weight", 1, H);
  }

  // biases = 0 (basic_transformer.py:113-115, 166-168 with config.biases = False): the four Linear layers of a block have no bias
  // PARAMETER -- nothing in the table, nothing in the state_dict -- but every kernel of the path takes a bias pointer: it points at
  // zeros behind n_trainable (never updated, never decayed, outside the clip-norm; the gradient sums written there are ignored)
  if (!c.biases) {
    auto hidden = [&](int n) {
      const int64_t off = e->n_params;
      e->n_params = (off + n + 63) & ~(int64_t)63;
      return off;
    };
    for (auto& x : e->xl) { x.attnb = hidden(3 * C); x.projb = hidden(C); x.fc1b = hidden(4 * C); x.fc2b = hidden(C); }
  }

  // --- shadow extras ---
  e->n_shadow = e->n_params;
  for (auto& x : e->xl) {
    x.attnT = add_shadow(e, (int64_t)C * 3 * C);
    x.projT = add_shadow(e, (int64_t)C * C);
    x.fc1T = add_shadow(e, (int64_t)C * 4 * C);
    x.fc2T = add_shadow(e, (int64_t)4 * C * C);
  }
  e->lmheadT = add_shadow(e, (int64_t)C * e->Vpad);
  for (auto& g : e->gl) {
    g.w1ab = add_shadow(e, (int64_t)2 * H * H);
    g.w1abT = add_shadow(e, (int64_t)H * 2 * H);
    g.e3T = add_shadow(e, (int64_t)H * H);
    g.n0T = add_shadow(e, (int64_t)2 * H * H);
    g.n3T = add_shadow(e, (int64_t)H * H);
    g.n0p = c.residual ? add_shadow(e, (int64_t)2 * H * H) : -1;
  }
  e->gd0T = add_shadow(e, (int64_t)H * H);
  e->gd3T = add_shadow(e, (int64_t)H * H);

  // fp8 mode: MXFP8 weight copies.  index 0..3 = natural [N, K] of attn / proj / fc1 / fc2 (forward products), 4..7 = the
  // transposed copies [K_in, N_out] (input-gradient products: the contraction runs over the layer's outputs)
  e->n_fp8 = 0;
  if (c.use_fp8) {
    for (auto& x : e->xl) {
      const int64_t rows[8] = {3 * C, C, 4 * C, C, C, C, C, 4 * C}, cols[8] = {C, C, C, 4 * C, 3 * C, C, 4 * C, C};
      for (int i = 0; i < 8; ++i) { x.q8[i] = e->n_fp8; e->n_fp8 += (rows[i] * cols[i] + 255) & ~(int64_t)255; }
      for (int i = 0; i < 8; ++i) { x.s8[i] = e->n_fp8; e->n_fp8 += (rows[i] * cols[i] / 32 + 255) & ~(int64_t)255; }
    }
  }

  // job table of the one-launch shadow refresh
  auto job = [&](int64_t src, int ld_src, int64_t dst, int ld_dst, int rows, int cols, int transpose) {
    e->tile_start.push_back(e->n_job_tiles);
    e->jobs.push_back({src, dst, ld_src, ld_dst, rows, cols, transpose});
    e->n_job_tiles += ((rows + 31) / 32) * ((cols + 31) / 32);
  };
  for (const XLayerP& w : e->xl) {
    job(w.attnw, C, w.attnT, 3 * C, 3 * C, C, 1);
    job(w.projw, C, w.projT, C, C, C, 1);
    job(w.fc1w, C, w.fc1T, 4 * C, 4 * C, C, 1);
    job(w.fc2w, 4 * C, w.fc2T, C, C, 4 * C, 1);
  }
  job(e->lmhead, C, e->lmheadT, e->Vpad, V, C, 1);
  for (const GLayerP& w : e->gl) {
    // W1 is [H, 2H+1]: receiver block W1a = cols [0,H), sender block W1b = cols [H,2H)
    job(w.e0w, 2 * H + 1, w.w1ab, H, H, H, 0);
    job(w.e0w + H, 2 * H + 1, w.w1ab + (int64_t)H * H, H, H, H, 0);
    job(w.e0w, 2 * H + 1, w.w1abT, 2 * H, H, H, 1);          // W1abT [H][2H]: T[k][n] = W1ab[n][k]
    job(w.e0w + H, 2 * H + 1, w.w1abT + H, 2 * H, H, H, 1);
    job(w.e3w, H, w.e3T, H, H, H, 1);
    job(w.n0w, 2 * H + (c.residual ? 28 : 0), w.n0T, H, H, 2 * H, 1);
    if (c.residual) job(w.n0w, 2 * H + 28, w.n0p, 2 * H, H, 2 * H, 0);   // (rows 2H + 28 apart are not 16-B aligned: the product reads a packed copy)
    job(w.n3w, H, w.n3T, H, H, H, 1);
  }
  job(e->gd0w, H, e->gd0T, H, H, H, 1);
  job(e->gd3w, H, e->gd3T, H, H, H, 1);
}

// ---- profiling wrapper ---------------------------------------------------------------------------------
constexpr int SITE_NONE = -2;   // launches outside the training step (never timed)
struct ProfScope {
  coati_engine* e;
  hipStream_t s;
  bool on;
  ProfScope(coati_engine* e_, int site, double flops, hipStream_t s_, double bytes = 0.0, double bytes_per_edge = 0.0, double flops_per_edge = 0.0)
      : e(e_), s(s_), on(false) {
    if (site >= 0 && !e->prof_paused && ((e->prof_mask >> site) & 1ull) && e->ev_used + 2 <= (int)e->ev.size()) {
      on = true;
      hipEventRecord(e->ev[e->ev_used], s);
      e->prof_flops += flops;
      e->prof_bytes += bytes;
      e->prof_bytes_e += bytes_per_edge;
      e->prof_flops_e += flops_per_edge;
    }
  }
  ~ProfScope() {
    if (on) {
      hipEventRecord(e->ev[e->ev_used + 1], s);
      e->ev_used += 2;
    }
  }
};

// ---- GEMM helpers -------------------------------------------------------------------------------------
// ---- row-wise products of MORE rows than one round of the fused kernels takes (batch 2048: ~ 100 000 packed rows per pass) -------------
// The 16-row-slab kernels (gemm_rb16.hip) and the one-round ring forms (gemm_ring.hip: the only ones with the LayerNorm backward in their
// write-out) serve 40 961 .. 57 344 rows; above 65 536 rows launch_gemm_nt falls back to the 32-row-slab and multi-round kernels and the
// step is 8.5 % slower PER ROW (batch_sweep, round 5).  Every one of these products is row-wise, so M rows run as n launches of the fast form on
// equal row ranges: the plan below (rows per launch; 0 = one launch as before) and gemm_rows() which advances every per-row operand.
// COATI_ROW_SPLIT=0 switches it off (A/B).
static bool row_split_on() {
  static const bool on = []() { const char* v = getenv("COATI_ROW_SPLIT"); return !(v && v[0] == '0'); }();
  return on;
}
static int row_split_plan(const GemmArgs& a, int a_f32, int epi) {
  if (!row_split_on() || a.m_dev != nullptr || a_f32 || a.M <= 65536 || a.q8_out != nullptr) return 0;
  if (!(a.K == 256 || a.N == 256)) return 0;
  if (epi != EPI_BF16 && epi != EPI_RES_F32 && epi != EPI_QKV_ROPE && epi != EPI_GELU_GRAD && epi != EPI_MUL_AUX && epi != EPI_CE_PARTIAL &&
      epi != EPI_CE_BWD && epi != EPI_LNBWD) return 0;
  // a launch starts on a 16-row slab and, in the padded layout of the rotary epilogue (row m sits at position m % rope_T), on a sequence
  // boundary
  long long unit = 16;
  if (epi == EPI_QKV_ROPE && a.rope_row_t == nullptr && a.rope_pos == nullptr && a.rope_T > 0) {
    long long g = unit, t = a.rope_T;
    while (t) { const long long r = g % t; g = t; t = r; }
    unit = unit / g * a.rope_T;
  }
  const int n = cdiv(a.M, 57344);
  const long long rows = (cdiv(a.M, n) + unit - 1) / unit * unit;
  if (rows <= 40960 || rows > 57344 || rows >= a.M) return 0;   // (40 960-row halves of a padded 1024 x 80 batch were tried: 33.3 vs 32.4 ms per step)
  if (epi == EPI_CE_PARTIAL) {   // the per-(row, tile) entries of every launch must have the layout the caller's buffer was sized for
    GemmArgs b = a;
    b.M = (int)rows;
    if (a.partial_tile != 64 || gemm_ce_tile_width(b) != 64) return 0;
  }
  return (int)rows;
}
static void advance_rows(GemmArgs& a, long long r0, int epi, int lnb_partial_rows_done) {
  const bool out32 = (epi == EPI_F32 || epi == EPI_RES_F32 || epi == EPI_ACC_F32 || epi == EPI_LNBWD);
  auto adv = [&](const void* q, long long bytes) -> const void* { return q ? reinterpret_cast<const char*>(q) + bytes : nullptr; };
  a.A = adv(a.A, r0 * a.lda * 2);
  a.C = const_cast<void*>(adv(a.C, r0 * a.ldc * (out32 ? 4 : 2)));
  const long long aux_in_b = (epi == EPI_RES_F32 || epi == EPI_LNBWD) ? 4 : (epi == EPI_MUL_AUX ? 1 : 2);
  const long long aux_out_b = (epi == EPI_GELU_GRAD) ? 1 : 2;
  a.aux_in = adv(a.aux_in, r0 * a.ld_aux * aux_in_b);
  a.aux_out = const_cast<void*>(adv(a.aux_out, r0 * a.ld_aux * aux_out_b));
  if (a.lse) a.lse += r0;
  if (a.target) a.target += r0;
  if (a.partial) a.partial += r0 * cdiv(a.N, a.partial_tile > 0 ? a.partial_tile : 128);
  if (a.rope_row_t) a.rope_row_t += r0;
  if (a.ln_x) { a.ln_x += r0 * a.ln_ldx; a.ln_mean += r0; a.ln_rstd += r0; }
  if (a.lnb_x) { a.lnb_x += r0 * a.lnb_ldx; a.lnb_mean += r0; a.lnb_rstd += r0; }
  if (a.lnb_partial) a.lnb_partial += (long long)lnb_partial_rows_done * 2 * a.N;
  if (a.chain_C) a.chain_C += r0 * a.chain_ldc;
}
// launch_gemm_nt on M rows, as n launches of `rows` rows where the plan says so.  lnb_rows (EPI_LNBWD): receives the number of partial rows
// the launches have left in lnb_partial.
static int gemm_rows(const GemmArgs& a, int a_f32, int epi, hipStream_t s, int* lnb_rows = nullptr) {
  const int rows = row_split_plan(a, a_f32, epi);
  if (rows == 0) {
    if (lnb_rows) { int nwg = 0; gemm_ring_lnbwd_supported(a, &nwg); *lnb_rows = nwg; }
    return launch_gemm_nt(a, a_f32, epi, s);
  }
  int done = 0;
  for (long long r0 = 0; r0 < a.M; r0 += rows) {
    GemmArgs b = a;
    b.M = (int)std::min<long long>(rows, a.M - r0);
    advance_rows(b, r0, epi, done);
    if (epi == EPI_LNBWD) {
      int nwg = 0;
      if (!gemm_ring_lnbwd_supported(b, &nwg)) { coati_set_error("gemm_rows: a row range of %d rows does not take the fused LayerNorm backward", b.M); return COATI_ESHAPE; }
      done += nwg;
    }
    COATI_TRY(launch_gemm_nt(b, a_f32, epi, s));
  }
  if (lnb_rows) *lnb_rows = done;
  return COATI_OK;
}
// true when every launch of the plan (or the single launch) takes the fused LayerNorm backward; *nwg = partial rows in total
static bool lnbwd_rows_supported(const GemmArgs& a, int* nwg) {
  const int rows = row_split_plan(a, 0, EPI_LNBWD);
  if (rows == 0) return gemm_ring_lnbwd_supported(a, nwg);
  int total = 0;
  for (long long r0 = 0; r0 < a.M; r0 += rows) {
    GemmArgs b = a;
    b.M = (int)std::min<long long>(rows, a.M - r0);
    int k = 0;
    if (!gemm_ring_lnbwd_supported(b, &k)) return false;
    total += k;
  }
  if (nwg) *nwg = total;
  return true;
}

// A/B switch of the point encoder's fused forward edge launch (COATI_GNN_EDGE_FUSED=0: the three launches)
static bool gnn_edge_fused_on() {
  static const bool on = []() { const char* v = getenv("COATI_GNN_EDGE_FUSED"); return !(v && v[0] == '0'); }();
  return on;
}
// A/B switch of the attention launch order (COATI_ATTN_LPT=0: sequences in batch order)
static bool lpt_on() {
  static const bool on = []() { const char* v = getenv("COATI_ATTN_LPT"); return !(v && v[0] == '0'); }();
  return on;
}
int gemm(coati_engine* e, int site, const void* A, int a_f32, int64_t lda, const bf16_t* Bm, int64_t ldb, int M,
         int N, int K, void* Cm, int64_t ldc, const float* bias, int epi, const void* aux_in, void* aux_out,
         int64_t ld_aux, hipStream_t s) {
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = Bm; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.C = Cm; a.ldc = ldc; a.bias = bias;
  a.aux_in = aux_in; a.aux_out = aux_out; a.ld_aux = ld_aux;
  // algorithmic bytes: A once, weights once, every output / extra operand once
  const bool out32 = (epi == EPI_F32 || epi == EPI_RES_F32 || epi == EPI_ACC_F32);
  double bytes = (double)M * K * (a_f32 ? 4 : 2) + (double)N * K * 2 + (double)M * N * (out32 ? 4 : 2);
  if (epi == EPI_RES_F32 || epi == EPI_ACC_F32) bytes += (double)M * N * 4;
  if (epi == EPI_GELU || epi == EPI_SILU || epi == EPI_DGELU || epi == EPI_DSILU) bytes += (double)M * N * 2;
  if (epi == EPI_GELU_GRAD || epi == EPI_MUL_AUX) bytes += (double)M * N;   // the saved NewGELU' is one byte per element
  ProfScope ps(e, site, 2.0 * M * N * K, s, bytes);
  return gemm_rows(a, a_f32, epi, s);
}
int wgrad(coati_engine* e, int site, const void* A, int a_f32, int64_t lda, const bf16_t* Bm, int64_t ldb, int M,
          int N, int K, float* dW, int64_t ldw, float* dbias, int n_out, hipStream_t s) {
  WgradArgs a;
  a.A = A; a.lda = lda; a.B = Bm; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.dW = dW; a.ldw = ldw; a.dbias = dbias;
  a.n_out = n_out;
  // algorithmic bytes: both activation operands once + the f32 gradient read-modify-write
  ProfScope ps(e, site, 2.0 * M * N * K, s, (double)M * N * (a_f32 ? 4 : 2) + (double)M * K * 2 + (double)N * K * 8);
  return launch_wgrad(a, a_f32, s);
}

// fp8 mode: one Linear product on MXFP8 -- quantise the bf16 A operand (rows of K) into the scratch, then the block-scaled fp8
// matrix-core GEMM against the layer's MXFP8 weight copy `wi` (XLayerP::q8 / s8 index) with the usual fused epilogue
// A == nullptr: the operand is already quantised in the second scratch set (emitted by the previous product's epilogue);
// emit: this product's epilogue leaves the MXFP8 copy of its bf16 output there for the next one
int gemm8(coati_engine* e, int site, const XLayerP& w, int wi, const bf16_t* A, int64_t lda, int M, int N, int K, GemmArgs a, int epi, hipStream_t s,
          bool emit = false) {
  COATI_CHECK_ARG(e->S8 && e->q8, "fp8 mode: coati_engine_bind_fp8 has not been called");
  const bool pre = A == nullptr;
  a.A = pre ? e->q8b : e->q8; a.lda = K; a.B = reinterpret_cast<const bf16_t*>(e->S8 + w.q8[wi]); a.ldb = K; a.M = M; a.N = N; a.K = K;
  if (emit) { a.q8_out = e->q8b; a.q8_scales = e->q8bs; a.ld_q8 = N; }
  const bool out32 = (epi == EPI_F32 || epi == EPI_RES_F32);
  double bytes = (pre ? 0.0 : (double)M * K * 3) + (double)M * K * (1.0 + 1.0 / 32) + (double)N * K * (1.0 + 1.0 / 32) + (double)M * N * (out32 ? 4 : 2);   // quantiser pass + product
  if (epi == EPI_RES_F32) bytes += (double)M * N * 4;
  if (epi == EPI_GELU_GRAD || epi == EPI_MUL_AUX) bytes += (double)M * N;
  if (emit) bytes += (double)M * N * (1.0 + 1.0 / 32);
  ProfScope ps(e, site, 2.0 * M * N * K, s, bytes);
  if (!pre) COATI_TRY(launch_quant_mx8(A, 0, lda, e->q8, K, e->q8s, M, K, s));
  return launch_gemm_mx8(a, pre ? e->q8bs : e->q8s, e->S8 + w.s8[wi], epi, s);
}

// ---- workspace carving -----------------------------------------------------------------------------------
void carve_pass(coati_engine* e, Arena& ar, XPass& p, int B_, int T_, int B, int T) {   // (B_, T_): the call's shape; (B, T) >= it: the capacity every buffer is sized for
  const int L = e->cfg.n_layer_xformer, C = e->cfg.n_hidden_xformer, nh = e->cfg.n_head;
  const size_t M = (size_t)B * T;
  p.B = B_; p.T = T_; p.M = B_ * T_; p.Mcap = (int)M;
  p.x.assign(L + 1, nullptr); p.xmid.assign(L, nullptr);
  p.mean1.assign(L, nullptr); p.rstd1.assign(L, nullptr); p.mean2.assign(L, nullptr); p.rstd2.assign(L, nullptr);
  p.lse.assign(L, nullptr);
  p.a1.assign(L, nullptr); p.qkv.assign(L, nullptr); p.y.assign(L, nullptr); p.a2.assign(L, nullptr);
  p.hpre.assign(L, nullptr); p.g.assign(L, nullptr);
  for (int l = 0; l <= L; ++l) p.x[l] = ar.take<float>(M * C);
  for (int l = 0; l < L; ++l) {
    p.xmid[l] = ar.take<float>(M * C);
    p.mean1[l] = ar.take<float>(M); p.rstd1[l] = ar.take<float>(M);
    p.mean2[l] = ar.take<float>(M); p.rstd2[l] = ar.take<float>(M);
    p.lse[l] = ar.take<float>((size_t)B * nh * T);
    p.a1[l] = ar.take<bf16_t>(M * C);
    p.qkv[l] = ar.take<bf16_t>(M * 3 * C);
    p.y[l] = ar.take<bf16_t>(M * C);
    p.a2[l] = ar.take<bf16_t>(M * C);
    p.hpre[l] = ar.take<unsigned char>(M * 4 * C);
    p.g[l] = ar.take<bf16_t>(M * 4 * C);
  }
  p.meanf = ar.take<float>(M); p.rstdf = ar.take<float>(M);
  p.xf32 = ar.take<float>(M * C);
  if (e->cfg.norm_embed) { p.x_emb = ar.take<float>(M * C); p.mean0 = ar.take<float>(M); p.rstd0 = ar.take<float>(M); }
  p.af = ar.take<bf16_t>(M * C);
  p.t_xmid = ar.take<float>((size_t)B * C); p.t_xL = ar.take<float>((size_t)B * C); p.t_xf = ar.take<float>((size_t)B * C);
  p.t_mean2 = ar.take<float>(B); p.t_rstd2 = ar.take<float>(B); p.t_meanf = ar.take<float>(B); p.t_rstdf = ar.take<float>(B);
  p.t_a2 = ar.take<bf16_t>((size_t)B * C); p.t_g = ar.take<bf16_t>((size_t)B * 4 * C); p.t_hpre = ar.take<unsigned char>((size_t)B * 4 * C);
  p.tail = false;
  p.packed = false;
  p.off = ar.take<int>((size_t)B + 1); p.ord = ar.take<int>((size_t)B); p.row_src = ar.take<int>(M); p.row_t = ar.take<int>(M); p.ypk = ar.take<long long>(M);
  p.grp = ar.take<int>((size_t)B + 2);
}

size_t carve(coati_engine* e, Arena& ar, int B_, int T1_, int T2_, int A_, int Bg) {
  const coati_config& c = e->cfg;
  // sizes below are CAPACITIES (coati_engine_reserve): the call's shape only sets the passes' B / T / M
  const int B = B_ > e->cap_B ? B_ : e->cap_B, T1 = T1_ > e->cap_T1 ? T1_ : e->cap_T1, T2 = T2_ > e->cap_T2 ? T2_ : e->cap_T2,
            A = A_ > e->cap_A ? A_ : e->cap_A;
  const int C = c.n_hidden_xformer, H = c.n_hidden_e3nn, E = c.n_embd_common, Lg = c.n_layer_e3gnn;
  const size_t Mmax = (size_t)B * (T1 > T2 ? T1 : T2), M2 = (size_t)B * T2;
  const size_t BA = (size_t)B * A, Me = BA * A;
  carve_pass(e, ar, e->p1, B_, T1_, B, T1);
  carve_pass(e, ar, e->p2, B_, T2_, B, T2);
  // heads
  e->hpoint = ar.take<float>((size_t)B * H); e->hp_ln = ar.take<float>((size_t)B * H);
  e->hp_mean = ar.take<float>(B); e->hp_rstd = ar.take<float>(B);
  e->hstop = ar.take<float>((size_t)B * C); e->hs_ln = ar.take<float>((size_t)B * C);
  e->hs_mean = ar.take<float>(B); e->hs_rstd = ar.take<float>(B);
  e->h_e3gnn = ar.take<float>((size_t)B * E); e->h_smiles = ar.take<float>((size_t)B * E);
  e->sa = ar.take<float>((size_t)B * E); e->sb = ar.take<float>((size_t)B * E);
  e->ptok = ar.take<float>((size_t)B * E); e->stok = ar.take<float>((size_t)B * E);
  e->cliptok = ar.take<float>((size_t)B * E);
  e->ones = ar.take<float>(B);
  e->stop_pos = ar.take<int>(B);
  e->err_flag = ar.take<int>(4);
  // lm head
  const int tiles_v = cdiv(c.n_tok, 64);   // one (max, sum) pair per 64 columns: either GEMM kernel fits
  e->ce_partial = ar.take<float2>(M2 * tiles_v);
  e->ce_lse = ar.take<float>(M2);
  e->dlogits = ar.take<bf16_t>(M2 * e->Vpad);
  // gnn
  e->g_h32.assign(Lg + 1, nullptr); e->g_rstd.assign(Lg + 1, nullptr);
  e->g_hcat.assign(Lg, nullptr); e->g_P.assign(Lg, nullptr); e->g_e1.assign(Lg, nullptr); e->g_s2.assign(Lg, nullptr);
  e->g_upre.assign(Lg, nullptr); e->g_t.assign(Lg, nullptr);
  for (int l = 0; l <= Lg; ++l) { e->g_h32[l] = ar.take<float>(BA * H); e->g_rstd[l] = ar.take<float>(BA); }
  for (int l = 0; l < Lg; ++l) {
    e->g_hcat[l] = ar.take<bf16_t>(BA * 2 * H); e->g_P[l] = ar.take<bf16_t>(BA * 2 * H);
    e->g_e1[l] = ar.take<bf16_t>(Me * H); e->g_s2[l] = ar.take<bf16_t>(Me * H);
    e->g_upre[l] = ar.take<bf16_t>(BA * H); e->g_t[l] = ar.take<bf16_t>(BA * H);
  }
  e->g_hfin16 = ar.take<bf16_t>(BA * H); e->g_dpre = ar.take<bf16_t>(BA * H); e->g_td = ar.take<bf16_t>(BA * H);
  e->g_mask = ar.take<float>(BA); e->g_d2 = ar.take<float>(Me); e->g_w = ar.take<float>(Me);
  e->g_seg = ar.take<int>(BA + 1); e->g_ne = ar.take<int>(4); e->g_ebj = ar.take<int>(Me); e->g_ebk = ar.take<int>(Me);
  e->g_erev = ar.take<int>(Me); e->g_pos = ar.take<int>(Me); e->g_ed2 = ar.take<float>(Me); e->g_ew = ar.take<float>(Me);
  e->g_o = ar.take<float>(BA * H); e->g_o2 = ar.take<float>(BA * H);
  // backward scratch
  e->DX = ar.take<float>(Mmax * C);
  e->DX16 = ar.take<bf16_t>(Mmax * C);
  e->g_DO16 = ar.take<bf16_t>(BA * H);
  e->dh4 = ar.take<bf16_t>(Mmax * 4 * C); e->da = ar.take<bf16_t>(Mmax * C); e->dyb = ar.take<bf16_t>(Mmax * C);
  e->dqkv = ar.take<bf16_t>(Mmax * 3 * C);
  e->attnD = ar.take<float>(Mmax * c.n_head);
  e->dcliptok = ar.take<float>((size_t)B * E); e->dptok = ar.take<float>((size_t)B * E); e->dstok = ar.take<float>((size_t)B * E);
  e->dsa = ar.take<float>((size_t)B * E); e->dsb = ar.take<float>((size_t)B * E);
  e->dhe = ar.take<float>((size_t)B * E); e->dhs = ar.take<float>((size_t)B * E);
  e->dhs_ln = ar.take<float>((size_t)B * C); e->dhstop = ar.take<float>((size_t)B * C);
  e->dhp_ln = ar.take<float>((size_t)B * H); e->dhpoint = ar.take<float>((size_t)B * H);
  e->g_DH = ar.take<float>(BA * H); e->g_DO = ar.take<float>(BA * H);
  e->g_do2 = ar.take<bf16_t>(BA * H); e->g_dtd = ar.take<bf16_t>(BA * H); e->g_du = ar.take<bf16_t>(BA * H);
  e->g_dmi = ar.take<bf16_t>(BA * H); e->g_ds2 = ar.take<bf16_t>(Me * H); e->g_dpre1 = ar.take<bf16_t>(Me * H);
  e->g_dP = ar.take<bf16_t>(BA * 2 * H);
  e->t_dx = ar.take<float>((size_t)B * C); e->t_dxa = ar.take<bf16_t>((size_t)B * C); e->t_dxb = ar.take<bf16_t>((size_t)B * C);
  e->t_dh4 = ar.take<bf16_t>((size_t)B * 4 * C); e->t_da = ar.take<bf16_t>((size_t)B * C);
  // deferred weight gradients: 9C bf16 per token and layer (6 GB at B*T = 81,920, L = 16: 288 GB of HBM make this free)
  {
    const int L = c.n_layer_xformer;
    e->wg_group = C % 128 == 0 && Mmax >= 4096;
    e->wg_tile = (C % 256 == 0 && 40LL * 4 * C < (1LL << 30)) ? 256 : 128;
    e->w_dh4.assign(L, nullptr); e->w_dxa.assign(L, nullptr); e->w_dxb.assign(L, nullptr); e->w_dqkv.assign(L, nullptr);
    if (e->wg_group) {
      for (int l = 0; l < L; ++l) {
        e->w_dh4[l] = ar.take<bf16_t>(Mmax * 4 * C); e->w_dxa[l] = ar.take<bf16_t>(Mmax * C);
        e->w_dxb[l] = ar.take<bf16_t>(Mmax * C); e->w_dqkv[l] = ar.take<bf16_t>(Mmax * 3 * C);
      }
      const int tpl = cdiv(3 * C, 128) * cdiv(C, 128) + cdiv(C, 128) * cdiv(C, 128) + 2 * cdiv(4 * C, 128) * cdiv(C, 128);
      e->wtab_cap = L * tpl;
      WgradTile* t = ar.take<WgradTile>((size_t)4 * e->wtab_cap);
      if (t != e->d_wtab || ar.base != e->wtab_ws) for (auto& k : e->wtab_key) k = coati_engine::WTabKey();
      e->d_wtab = t;
      e->wtab_ws = ar.base;
    }
    // The cached tile tables live INSIDE the workspace: any carve for another shape (an encode call, a smaller last batch, a
    // shape that does not take the grouped launch) lays other buffers over them, so the cache is only valid while the very
    // same (B, T1, T2, A) is carved again
    const long long csig = (((long long)B * 1000003 + T1) * 1000003 + T2) * 1000003 + A;
    if (!e->wg_group || csig != e->carve_sig) for (auto& k : e->wtab_key) k = coati_engine::WTabKey();
    {
      e->gnn_wg_group = H % 128 == 0 && BA >= 4096 && Lg > 0;
      e->gl_DO16.assign(Lg, nullptr); e->gl_du.assign(Lg, nullptr); e->gl_dP.assign(Lg, nullptr);
      if (e->gnn_wg_group) {
        for (int l = 0; l < Lg; ++l) { e->gl_DO16[l] = ar.take<bf16_t>(BA * H); e->gl_du[l] = ar.take<bf16_t>(BA * H); e->gl_dP[l] = ar.take<bf16_t>(BA * 2 * H); }
        e->gtab_cap = 4 * (2 + Lg * 5) * cdiv(H, 128) * cdiv(H, 128);   // 4 slices x (2 decoder + per layer 3 of H x H and one of H x 2H) tiles
        WgradTile* t = ar.take<WgradTile>((size_t)e->gtab_cap);
        if (t != e->d_gtab || ar.base != e->gtab_ws || csig != e->carve_sig) e->gtab_sig = -1;
        e->d_gtab = t;
        e->gtab_ws = ar.base;
      } else {
        e->gtab_sig = -1;
      }
    }
    e->carve_sig = csig;
  }
  if (c.use_fp8) {   // the quantised A operand of one product (at most 4 C wide) + its block scales
    e->q8 = ar.take<unsigned char>(Mmax * 4 * C);
    e->q8s = ar.take<unsigned char>(Mmax * 4 * C / 32);
    e->q8b = ar.take<unsigned char>(Mmax * 4 * C);
    e->q8bs = ar.take<unsigned char>(Mmax * 4 * C / 32);
  }
  e->opt_partial = ar.take<float>(1024);
  e->ln_partial = ar.take<float>((size_t)COATI_LN_PARTIAL_ROWS * 2 * (C > H ? C : H));
  e->ln_part_x = ar.take<float>((size_t)(2 * c.n_layer_xformer + 1) * COATI_LN_PARTIAL_ROWS * 2 * C);
  {
    ShadowJob* dj = ar.take<ShadowJob>(e->jobs.size());
    int* dt = ar.take<int>(e->tile_start.size());
    if (dj != e->d_jobs || dt != e->d_tile_start) e->jobs_uploaded = false;
    e->d_jobs = dj; e->d_tile_start = dt;
  }
  // LAST region: InfoNCE logits of the local rows against the GLOBAL batch, [B, Bg] f32 x 2 (Bg = world_size * B at
  // workspace-sizing time).  coati_engine_forward does not know Bg: it hands this region whatever the caller's
  // workspace holds beyond everything above.
  e->nce_cap = (size_t)2 * B_ * (Bg > B_ ? Bg : B_);
  e->nce = ar.take<float>(e->nce_cap);
  return (ar.off + 255) & ~(size_t)255;
}

// ---- transformer pass ---------------------------------------------------------------------------------
int xformer_fwd(coati_engine* e, XPass& p, const float* injection, hipStream_t s) {
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer, L = c.n_layer_xformer, M = p.M;
  {
    ProfScope ps(e, SITE_EMBED, 0, s);
    if (c.norm_embed) {
      // x = LayerNorm(tok_emb[idx]); x[idx == [UNK]] = injection (smiles_xformer.py:442-448: the injection replaces the NORMALISED rows)
      COATI_TRY(launch_embed_fwd(p.idx, e->P + e->tok_emb, nullptr, c.unk_token, p.x_emb, p.B, p.T, C, c.n_tok, s, p.packed ? p.row_src : nullptr, M));
      COATI_TRY(launch_layernorm_fwd(p.x_emb, C, e->P + e->emb_lnw, e->P + e->emb_lnb, nullptr, 0, p.x[0], C, p.mean0, p.rstd0, M, C, s));
      if (injection != nullptr)
        COATI_TRY(launch_embed_fwd(p.idx, nullptr, injection, c.unk_token, p.x[0], p.B, p.T, C, c.n_tok, s, p.packed ? p.row_src : nullptr, M, 1));
    } else {
      COATI_TRY(launch_embed_fwd(p.idx, e->P + e->tok_emb, injection, c.unk_token, p.x[0], p.B, p.T, C, c.n_tok, s, p.packed ? p.row_src : nullptr, M));
    }
  }
  // The attention half of every block as ONE launch (attn_block.hip): ln_1 -> c_attn -> RoPE -> causal attention -> c_proj -> + x
  // with qkv and y written once and never read back (d = 256, 16 heads, sequences of <= 128 rows); its work list -- groups of
  // whole sequences -- is built on the device once per pass
#ifdef COATI_EXPERIMENTAL
  const bool ab = !c.use_fp8 && attn_block_fwd_supported(p.B, p.T, C, c.n_head);
  if (ab) COATI_TRY(launch_attn_groups(p.packed ? p.off : nullptr, p.B, p.T, p.grp, s));
#else
  constexpr bool ab = false;
#endif
  for (int l = 0; l < L; ++l) {
    const XLayerP& w = e->xl[l];
#ifdef COATI_EXPERIMENTAL
    if (ab) {
      AttnBlockArgs a;
      a.x = p.x[l]; a.xmid = p.xmid[l]; a.ln_g = e->P + w.ln1w; a.ln_b = e->P + w.ln1b; a.mean = p.mean1[l]; a.rstd = p.rstd1[l];
      a.a1 = p.a1[l]; a.Wqkv = e->S + w.attnw; a.bqkv = e->P + w.attnb; a.Wproj = e->S + w.projw; a.bproj = e->P + w.projb;
      a.qkv = p.qkv[l]; a.y = p.y[l]; a.lse = p.lse[l]; a.cos_t = e->cos_t; a.sin_t = e->sin_t; a.row_src = p.packed ? p.row_src : nullptr;
      a.grp = p.grp; a.Tl = p.T; a.M = M;
      // algorithmic bytes: x in, a1 + qkv + y + xmid + lse + statistics out, both weights
      ProfScope ps(e, SITE_ATTN_BLOCK_FWD, 2.0 * M * 4 * C * C + 4.0 * M * (double)p.T * C, s,
                   (double)M * C * (4 + 2 + 6 + 2 + 4) + (double)M * (c.n_head * 4 + 8) + 4.0 * C * C * 2);
      COATI_TRY(launch_attn_block_fwd(a, s));
    }
#endif
    if (c.use_fp8) {
      // MXFP8 products (BASELINE.json configs[4]); LayerNorm, attention, residual stream and saved tensors as in the bf16 path
      GemmArgs a;
      {
        ProfScope ps(e, SITE_LN_FWD, 0, s, (double)M * C * 6 + (double)M * 8);
        COATI_TRY(launch_layernorm_fwd(p.x[l], C, e->P + w.ln1w, e->P + w.ln1b, p.a1[l], C, nullptr, 0, p.mean1[l], p.rstd1[l], M, C, s));
      }
      memset(&a, 0, sizeof(a));
      a.C = p.qkv[l]; a.ldc = 3 * C; a.bias = e->P + w.attnb; a.rope_cos = e->cos_t; a.rope_sin = e->sin_t; a.rope_T = p.T; a.rope_C = C;
      a.rope_hs = C / c.n_head; a.rope_row_t = p.packed ? p.row_t : nullptr;
      COATI_TRY(gemm8(e, SITE_QKV_FWD, w, 0, p.a1[l], C, M, 3 * C, C, a, EPI_QKV_ROPE, s));
      {
        ProfScope ps(e, SITE_ATTN_FWD, 4.0 * M * (double)p.T * C, s, (double)M * 4 * C * 2 + (double)M * c.n_head * 4);
        COATI_TRY(launch_attn_fwd(p.qkv[l], p.y[l], p.lse[l], p.B, p.T, c.n_head, C / c.n_head, s, p.packed ? p.off : nullptr, p.packed && lpt_on() ? p.ord : nullptr));
      }
      memset(&a, 0, sizeof(a));
      a.C = p.xmid[l]; a.ldc = C; a.bias = e->P + w.projb; a.aux_in = p.x[l]; a.ld_aux = C;
      COATI_TRY(gemm8(e, SITE_PROJ_FWD, w, 1, p.y[l], C, M, C, C, a, EPI_RES_F32, s));
      {
        ProfScope ps(e, SITE_LN_FWD, 0, s, (double)M * C * 6 + (double)M * 8);
        COATI_TRY(launch_layernorm_fwd(p.xmid[l], C, e->P + w.ln2w, e->P + w.ln2b, p.a2[l], C, nullptr, 0, p.mean2[l], p.rstd2[l], M, C, s));
      }
      memset(&a, 0, sizeof(a));
      a.C = p.g[l]; a.ldc = 4 * C; a.bias = e->P + w.fc1b; a.aux_out = p.hpre[l]; a.ld_aux = 4 * C;
      COATI_TRY(gemm8(e, SITE_FC1_FWD, w, 2, p.a2[l], C, M, 4 * C, C, a, EPI_GELU_GRAD, s, true));   // + g as MXFP8 for FC2
      memset(&a, 0, sizeof(a));
      a.C = p.x[l + 1]; a.ldc = C; a.bias = e->P + w.fc2b; a.aux_in = p.xmid[l]; a.ld_aux = C;
      COATI_TRY(gemm8(e, SITE_FC2_FWD, w, 3, nullptr, 4 * C, M, C, 4 * C, a, EPI_RES_F32, s));
      continue;
    }
    if (!ab) {
      // QKV projection with RoPE applied to the q,k blocks in the epilogue (saved qkv holds the ROTATED q,k).  Where the
      // row-block kernel takes the product, ln_1 is evaluated inside its operand load (x f32 in; a1, mean, rstd out): no
      // LayerNorm launch, no second trip of the normalised rows through HBM
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.A = p.a1[l]; a.lda = C; a.B = e->S + w.attnw; a.ldb = C; a.M = M; a.N = 3 * C; a.K = C; a.C = p.qkv[l]; a.ldc = 3 * C;
      a.bias = e->P + w.attnb; a.rope_cos = e->cos_t; a.rope_sin = e->sin_t; a.rope_T = p.T; a.rope_C = C; a.rope_hs = C / c.n_head;
      a.rope_row_t = p.packed ? p.row_t : nullptr;
      const bool fuse = gemm_rb256_ln_fusable(a, EPI_QKV_ROPE);
      if (fuse) {
        a.ln_x = p.x[l]; a.ln_ldx = C; a.ln_gamma = e->P + w.ln1w; a.ln_beta = e->P + w.ln1b; a.ln_mean = p.mean1[l]; a.ln_rstd = p.rstd1[l];
      } else {
        ProfScope ps(e, SITE_LN_FWD, 0, s, (double)M * C * 6 + (double)M * 8);
        COATI_TRY(launch_layernorm_fwd(p.x[l], C, e->P + w.ln1w, e->P + w.ln1b, p.a1[l], C, nullptr, 0, p.mean1[l], p.rstd1[l], M, C, s));
      }
      ProfScope ps(e, SITE_QKV_FWD, 2.0 * M * 3 * C * C, s, (double)M * C * (fuse ? 6 : 2) + 3.0 * C * C * 2 + (double)M * 3 * C * 2);
      COATI_TRY(gemm_rows(a, 0, EPI_QKV_ROPE, s));
    }
    if (!ab) {
      {
        ProfScope ps(e, SITE_ATTN_FWD, 4.0 * M * (double)p.T * C, s, (double)M * 4 * C * 2 + (double)M * c.n_head * 4);   // qkv in, y + lse out
        COATI_TRY(launch_attn_fwd(p.qkv[l], p.y[l], p.lse[l], p.B, p.T, c.n_head, C / c.n_head, s, p.packed ? p.off : nullptr, p.packed && lpt_on() ? p.ord : nullptr));
      }
      COATI_TRY(gemm(e, SITE_PROJ_FWD, p.y[l], 0, C, e->S + w.projw, C, M, C, C, p.xmid[l], C, e->P + w.projb, EPI_RES_F32, p.x[l], nullptr, C, s));
    }
    if (p.tail && l == L - 1) {
      // the [STOP] rows alone from here on: gather, ln_2, MLP + residual, ln_f (B rows instead of M)
      const int B = p.B;
      COATI_TRY(launch_gather_rows(p.xmid[l], e->stop_pos, p.t_xmid, B, p.T, C, s, p.packed ? p.off : nullptr));
      {
        ProfScope ps(e, SITE_XF_TAIL, 0, s, (double)B * C * 6 + (double)B * 8);
        COATI_TRY(launch_layernorm_fwd(p.t_xmid, C, e->P + w.ln2w, e->P + w.ln2b, p.t_a2, C, nullptr, 0, p.t_mean2, p.t_rstd2, B, C, s));
      }
      COATI_TRY(gemm(e, SITE_XF_TAIL, p.t_a2, 0, C, e->S + w.fc1w, C, B, 4 * C, C, p.t_g, 4 * C, e->P + w.fc1b, EPI_GELU_GRAD, nullptr, p.t_hpre, 4 * C, s));
      COATI_TRY(gemm(e, SITE_XF_TAIL, p.t_g, 0, 4 * C, e->S + w.fc2w, 4 * C, B, C, 4 * C, p.t_xL, C, e->P + w.fc2b, EPI_RES_F32, p.t_xmid, nullptr, C, s));
      ProfScope ps(e, SITE_XF_TAIL, 0, s, (double)B * C * 8 + (double)B * 8);
      return launch_layernorm_fwd(p.t_xL, C, e->P + e->lnfw, e->P + e->lnfb, nullptr, 0, p.t_xf, C, p.t_meanf, p.t_rstdf, B, C, s);
    }
    {
      // hpre holds NewGELU'(pre-activation), not the pre-activation: the backward multiplies instead of re-evaluating the
      // sigmoid (the activation epilogues are VALU-bound: 2 quarter-rate transcendentals per element).  ln_2 is fused into
      // the operand load like ln_1 above.
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.A = p.a2[l]; a.lda = C; a.B = e->S + w.fc1w; a.ldb = C; a.M = M; a.N = 4 * C; a.K = C; a.C = p.g[l]; a.ldc = 4 * C;
      a.bias = e->P + w.fc1b; a.aux_out = p.hpre[l]; a.ld_aux = 4 * C;
      const bool fuse = gemm_rb256_ln_fusable(a, EPI_GELU_GRAD);
      if (fuse) {
        a.ln_x = p.xmid[l]; a.ln_ldx = C; a.ln_gamma = e->P + w.ln2w; a.ln_beta = e->P + w.ln2b; a.ln_mean = p.mean2[l]; a.ln_rstd = p.rstd2[l];
      } else {
        ProfScope ps(e, SITE_LN_FWD, 0, s, (double)M * C * 6 + (double)M * 8);
        COATI_TRY(launch_layernorm_fwd(p.xmid[l], C, e->P + w.ln2w, e->P + w.ln2b, p.a2[l], C, nullptr, 0, p.mean2[l], p.rstd2[l], M, C, s));
      }
      ProfScope ps(e, SITE_FC1_FWD, 2.0 * M * 4 * C * C, s, (double)M * C * (fuse ? 6 : 2) + 4.0 * C * C * 2 + (double)M * 4 * C * 3);
      COATI_TRY(gemm_rows(a, 0, EPI_GELU_GRAD, s));
    }
    COATI_TRY(gemm(e, SITE_FC2_FWD, p.g[l], 0, 4 * C, e->S + w.fc2w, 4 * C, M, C, 4 * C, p.x[l + 1], C, e->P + w.fc2b, EPI_RES_F32, p.xmid[l], nullptr, C, s));
  }
  // the f32 copy of ln_f's output is what the [STOP] rows are gathered from: encoder pass only (the decoder pass feeds lm_head: bf16)
  float* const y32 = (&p == &e->p1) ? p.xf32 : nullptr;
  ProfScope ps(e, SITE_LN_FWD, 0, s, (double)M * C * (y32 ? 8 : 6) + (double)M * 8);
  return launch_layernorm_fwd(p.x[L], C, e->P + e->lnfw, e->P + e->lnfb, y32 ? nullptr : p.af, C, y32, C, p.meanf, p.rstdf, M, C, s);
}

// Asynchronous upload of a host-built table: the entries are copied into a PINNED staging buffer owned by the engine (one per slot;
// before a slot's buffer is rewritten the host waits for the event of its previous upload -- four misses back, i.e. never in
// practice) and hipMemcpyAsync runs from there in stream order.  No hipStreamSynchronize: a cache miss (another batch shape) used to
// stall the host until the device had drained the step.
int upload_table(coati_engine* e, int hslot, const std::vector<WgradTile>& tab, WgradTile* dst, hipStream_t s, const char* what) {
  const size_t n = tab.size();
  if (e->h_tab_ev[hslot] == nullptr && hipEventCreateWithFlags(&e->h_tab_ev[hslot], hipEventDisableTiming) != hipSuccess) {
    coati_set_error("%s: event creation failed", what);
    return COATI_EHIP;
  } else if (e->h_tab[hslot] != nullptr && hipEventSynchronize(e->h_tab_ev[hslot]) != hipSuccess) {
    coati_set_error("%s: waiting for the previous upload failed", what);
    return COATI_EHIP;
  }
  if (e->h_tab_cap[hslot] < n) {
    if (e->h_tab[hslot] != nullptr) hipHostFree(e->h_tab[hslot]);
    e->h_tab[hslot] = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&e->h_tab[hslot]), n * sizeof(WgradTile), hipHostMallocDefault) != hipSuccess) {
      coati_set_error("%s: pinned staging allocation failed", what);
      return COATI_EHIP;
    }
    e->h_tab_cap[hslot] = n;
  }
  memcpy(e->h_tab[hslot], tab.data(), n * sizeof(WgradTile));
  if (hipMemcpyAsync(dst, e->h_tab[hslot], n * sizeof(WgradTile), hipMemcpyHostToDevice, s) != hipSuccess || hipEventRecord(e->h_tab_ev[hslot], s) != hipSuccess) {
    coati_set_error("%s: table upload failed", what);
    return COATI_EHIP;
  }
  return COATI_OK;
}

// One launch for the 4 (l_hi - l_lo) weight gradients of a layer range (bias gradients included): the tile table is built
// once per (pass, range, shape) and cached in the workspace.
int xformer_wgrad_group(coati_engine* e, XPass& p, int l_lo, int l_hi, hipStream_t s) {
  // the table is built for the padded row count of the pass (it only fixes operand addresses and tile indices); the rows a
  // launch really streams (fewer with packed rows, different every batch) are a kernel argument
  const int C = e->cfg.n_hidden_xformer, M = p.M, Mtab = p.Mcap;
  const long long sig = e->carve_sig * 2 + (p.tail ? 1 : 0);      // the capacity layout: a batch with another T / row count reuses the table
  int slot = -1;
  for (int i = 0; i < 4; ++i)
    if (e->wtab_key[i].pass == &p && e->wtab_key[i].lo == l_lo && e->wtab_key[i].hi == l_hi && e->wtab_key[i].M == Mtab && e->wtab_key[i].sig == sig) slot = i;
  double flops = 0.0, bytes = 0.0;
  for (int l = l_lo; l < l_hi; ++l) {
    flops += 2.0 * M * (3.0 + 1.0 + 4.0 + 4.0) * C * C;
    bytes += (double)M * (3 + 1 + 1 + 1 + 4 + 1 + 1 + 4) * C * 2 + 12.0 * C * C * 8;   // every operand once + the f32 gradient read-modify-write
  }
  if (slot < 0) {
    std::vector<WgradTile> tab;
    auto add = [&](const bf16_t* A, int lda, const bf16_t* B, int ldb, int N, int K, int64_t w_off, int64_t b_off) -> int {
      WgradArgs a;
      a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.M = Mtab; a.N = N; a.K = K; a.dW = e->G + w_off; a.ldw = K; a.dbias = e->G + b_off; a.n_out = 0;
      return wgrad_table_append(tab, a, e->wg_tile);
    };
    for (int l = l_hi - 1; l >= l_lo; --l) {
      const XLayerP& w = e->xl[l];
      COATI_TRY(add(e->w_dqkv[l], 3 * C, p.a1[l], C, 3 * C, C, w.attnw, w.attnb));
      COATI_TRY(add(e->w_dxb[l], C, p.y[l], C, C, C, w.projw, w.projb));
      if (p.tail && l == e->cfg.n_layer_xformer - 1) continue;   // the tail layer's MLP ran on the [STOP] rows only: its two gradients are small launches in xformer_bwd
      COATI_TRY(add(e->w_dh4[l], 4 * C, p.a2[l], C, 4 * C, C, w.fc1w, w.fc1b));
      COATI_TRY(add(e->w_dxa[l], C, p.g[l], 4 * C, C, 4 * C, w.fc2w, w.fc2b));
    }
    COATI_CHECK_ARG((int)tab.size() <= e->wtab_cap, "wgrad group: table overflow (%zu > %d)", tab.size(), e->wtab_cap);
    slot = e->wtab_rr++ & 3;
    COATI_TRY(upload_table(e, slot, tab, e->d_wtab + (size_t)slot * e->wtab_cap, s, "wgrad group"));
    coati_engine::WTabKey k;
    k.pass = &p; k.lo = l_lo; k.hi = l_hi; k.M = Mtab; k.n = (int)tab.size(); k.sig = sig;
    e->wtab_key[slot] = k;
  }
  ProfScope ps(e, SITE_XF_WGRAD, flops, s, bytes);
  return launch_wgrad_table(e->d_wtab + (size_t)slot * e->wtab_cap, e->wtab_key[slot].n, s, e->wg_tile, M);
}

// dyf: gradient w.r.t. ln_f output, bf16 (decoder pass) or f32 (encoder pass)
// layers [l_lo, l_hi) only, descending; the ln_f backward belongs to l_hi == L, the embedding backward to l_lo == 0
// lmh_dlogits (decoder pass): d logits [M, Vpad] bf16 -- dyf (= e->da) has NOT been computed yet, xformer_bwd runs the lm_head's
// input-gradient product itself (with ln_f's backward fused into it at packed-batch sizes)
int xformer_bwd(coati_engine* e, XPass& p, const void* dyf, int dyf_f32, float* dinjection, hipStream_t s, int l_hi = -1, int l_lo = 0,
                const bf16_t* lmh_dlogits = nullptr) {
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer, L = c.n_layer_xformer, M = p.M;
  if (l_hi < 0) l_hi = L;
  float* DX = e->DX;
  // the dgamma / dbeta partial sums of the pass's 2L + 1 LayerNorms are added up by ONE launch at the end
  const bool defer = 2 * L + 1 <= COATI_LN_MAX_SLOTS;
  const long long slot_stride = (long long)COATI_LN_PARTIAL_ROWS * 2 * C;
  LnFinishBatch fin;
  memset(&fin, 0, sizeof(fin));
  int nblk = 0;
  auto ln_bwd = [&](const void* dy, int dy_f32, const float* x, const float* mean, const float* rstd, const float* gamma,
                    const float* dres, size_t goff, size_t boff, bf16_t* dx16) -> int {
    if (!defer) return launch_layernorm_bwd(dy, dy_f32, C, x, C, 0, mean, rstd, gamma, dres, DX, dx16, e->G + goff, e->G + boff, e->ln_partial, M, C, s);
    fin.dg_off[fin.n] = (long long)goff;
    fin.db_off[fin.n] = (long long)boff;
    return launch_layernorm_bwd_deferred(dy, dy_f32, C, x, C, 0, mean, rstd, gamma, dres, DX, dx16, e->ln_part_x + (fin.n++) * slot_stride, &nblk, M, C, s);
  };
  // An input-gradient product whose result is the gradient w.r.t. a LayerNorm's OUTPUT (fc1 -> ln_2, c_attn -> ln_1) with that
  // LayerNorm's backward in the product's write-out (gemm_ring.hip EPI_LNBWD): dy never visits HBM, one launch instead of two.
  // Returns 1 when the fused launch ran, 0 when the shape does not take it (the caller then runs the two launches), < 0 on error.
  // chainW / chainC: the product that consumes the LayerNorm backward's bf16 output (c_proj's input gradient behind ln_2) in the
  // same launch
  auto dgrad_lnbwd = [&](int site, const bf16_t* dY, int K, const bf16_t* WT, const float* x, const float* mean, const float* rstd,
                         const float* gamma, size_t goff, size_t boff, bf16_t* dx16, const bf16_t* chainW = nullptr,
                         bf16_t* chainC = nullptr, bool has_dres = true) -> int {
    if (!defer || c.use_fp8) return 0;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = dY; a.lda = K; a.B = WT; a.ldb = K; a.M = M; a.N = C; a.K = K; a.C = DX; a.ldc = C; a.aux_in = has_dres ? DX : nullptr; a.ld_aux = C; a.aux_out = dx16;
    a.lnb_x = x; a.lnb_ldx = C; a.lnb_mean = mean; a.lnb_rstd = rstd; a.lnb_gamma = gamma;
    if (chainW != nullptr && dx16 != nullptr) { a.chain_W = chainW; a.chain_ldw = C; a.chain_C = chainC; a.chain_ldc = C; }
    int nwg = 0;
    if (!lnbwd_rows_supported(a, &nwg) || nwg > COATI_LN_PARTIAL_ROWS) return 0;
    a.lnb_partial = e->ln_part_x + fin.n * slot_stride;
    fin.dg_off[fin.n] = (long long)goff;
    fin.db_off[fin.n] = (long long)boff;
    fin.nblk[fin.n] = nwg;
    ++fin.n;
    // algorithmic bytes: dY + weight in; x, dres in; dx (f32) + its bf16 copy out
    ProfScope ps(e, site, 2.0 * M * C * K + (a.chain_W ? 2.0 * M * C * C : 0.0), s,
                 (double)M * K * 2 + (double)C * K * 2 + (double)M * C * (4 + (has_dres ? 4 : 0) + 4 + (dx16 ? 2 : 0)) + (a.chain_W ? (double)M * C * 2 + (double)C * C * 2 : 0.0));
    const int rc = gemm_rows(a, 0, EPI_LNBWD, s);
    return rc == COATI_OK ? 1 : rc;
  };
  // Weight gradients: immediately, one launch per Linear (small shapes), or deferred to ONE grouped launch at the end of
  // this call (grouped = every activation gradient of the layer range stays alive in its own buffer).
  const bool grp = e->wg_group && M >= 4096;
  if (l_hi == L && p.tail) {
    // ln_f backward on the B tail rows (dyf = [B, C]); not deferred: its partial sums have their own row count
    ProfScope ps(e, SITE_XF_TAIL, 0, s, (double)p.B * C * ((dyf_f32 ? 4 : 2) + 4 + 4 + 2));
    COATI_TRY(launch_layernorm_bwd(dyf, dyf_f32, C, p.t_xL, C, 0, p.t_meanf, p.t_rstdf, e->P + e->lnfw, nullptr, e->t_dx, e->t_dxa, e->G + e->lnfw, e->G + e->lnfb, e->ln_partial, p.B, C, s));
  } else if (l_hi == L) {
    // decoder pass: dyf is the lm_head's input gradient, not yet computed (lmh_dlogits != null): the product runs here, with ln_f's
    // backward in its write-out where the shape takes it
    int lnf_fused = 0;
    if (lmh_dlogits != nullptr) {
      lnf_fused = dgrad_lnbwd(SITE_LMHEAD_DGRAD, lmh_dlogits, e->Vpad, e->S + e->lmheadT, p.x[L], p.meanf, p.rstdf, e->P + e->lnfw, e->lnfw, e->lnfb,
                              grp ? e->w_dxa[L - 1] : e->DX16, nullptr, nullptr, false);
      if (lnf_fused < 0) return lnf_fused;
      if (!lnf_fused) COATI_TRY(gemm(e, SITE_LMHEAD_DGRAD, lmh_dlogits, 0, e->Vpad, e->S + e->lmheadT, e->Vpad, M, C, e->Vpad, e->da, C, nullptr, EPI_BF16, nullptr, nullptr, 0, s));
    }
    if (!lnf_fused) {
      ProfScope ps(e, SITE_LN_BWD, 0, s, (double)M * C * ((dyf_f32 ? 4 : 2) + 4 + 4 + 2));
      COATI_TRY(ln_bwd(dyf, dyf_f32, p.x[L], p.meanf, p.rstdf, e->P + e->lnfw, nullptr, e->lnfw, e->lnfb, grp ? e->w_dxa[L - 1] : e->DX16));
    }
  }
  for (int l = l_hi - 1; l >= l_lo; --l) {
    const XLayerP& w = e->xl[l];
    bf16_t* const dxa = grp ? e->w_dxa[l] : e->DX16;      // d x[l+1]
    bf16_t* const dxb = grp ? e->w_dxb[l] : e->DX16;      // d xmid[l]
    bf16_t* const dh4 = grp ? e->w_dh4[l] : e->dh4;
    bf16_t* const dqkv = grp ? e->w_dqkv[l] : e->dqkv;
    bf16_t* const dx_out = l == 0 ? nullptr : (grp ? e->w_dxa[l - 1] : e->DX16);   // d x[l] (bf16) for the layer below; below layer 0 the embedding backward reads the f32 stream
    int ln2_fused = 0, ln1_fused = 0;
    if (p.tail && l == L - 1) {
      // the last layer's MLP and ln_2 exist on the B [STOP] rows only (XPass::tail): their backward on those rows, the two weight
      // gradients as small launches of their own (the grouped table leaves them out), then the residual-stream gradient is
      // spread back over the M rows -- zero everywhere else -- for the attention half of the layer
      const int Bt = p.B;
      COATI_TRY(gemm(e, SITE_XF_TAIL, e->t_dxa, 0, C, e->S + w.fc2T, C, Bt, 4 * C, C, e->t_dh4, 4 * C, nullptr, EPI_MUL_AUX, p.t_hpre, nullptr, 4 * C, s));
      COATI_TRY(gemm(e, SITE_XF_TAIL, e->t_dh4, 0, 4 * C, e->S + w.fc1T, 4 * C, Bt, C, 4 * C, e->t_da, C, nullptr, EPI_BF16, nullptr, nullptr, 0, s));
      COATI_TRY(wgrad(e, SITE_XF_TAIL, e->t_dh4, 0, 4 * C, p.t_a2, C, Bt, 4 * C, C, e->G + w.fc1w, C, e->G + w.fc1b, 0, s));
      COATI_TRY(wgrad(e, SITE_XF_TAIL, e->t_dxa, 0, C, p.t_g, 4 * C, Bt, C, 4 * C, e->G + w.fc2w, 4 * C, e->G + w.fc2b, 0, s));
      {
        ProfScope ps(e, SITE_XF_TAIL, 0, s, (double)Bt * C * (2 + 4 + 4 + 4 + 2));
        COATI_TRY(launch_layernorm_bwd(e->t_da, 0, C, p.t_xmid, C, 0, p.t_mean2, p.t_rstd2, e->P + w.ln2w, e->t_dx, e->t_dx, e->t_dxb, e->G + w.ln2w, e->G + w.ln2b, e->ln_partial, Bt, C, s));
        if (hipMemsetAsync(DX, 0, (size_t)M * C * sizeof(float), s) != hipSuccess || hipMemsetAsync(dxb, 0, (size_t)M * C * sizeof(bf16_t), s) != hipSuccess) {
          coati_set_error("xformer_bwd: hipMemsetAsync failed");
          return COATI_EHIP;
        }
        COATI_TRY(launch_scatter_rows_add(e->t_dx, e->stop_pos, DX, Bt, p.T, C, s, p.packed ? p.off : nullptr));
        COATI_TRY(launch_scatter_rows_bf16(e->t_dxb, e->stop_pos, dxb, Bt, p.T, C, s, p.packed ? p.off : nullptr));
      }
    } else {
      // x[l+1] = xmid + g W2^T + b2
      if (c.use_fp8) {   // input gradients on MXFP8: the gradient rows are quantised like the activations (e4m3, block of 32 along k)
        GemmArgs a;
        memset(&a, 0, sizeof(a));
        a.C = dh4; a.ldc = 4 * C; a.aux_in = p.hpre[l]; a.ld_aux = 4 * C;
        COATI_TRY(gemm8(e, SITE_FC2_DGRAD, w, 7, dxa, C, M, 4 * C, C, a, EPI_MUL_AUX, s, true));   // + d hidden as MXFP8 for the FC1 input gradient
        memset(&a, 0, sizeof(a));
        a.C = e->da; a.ldc = C;
        COATI_TRY(gemm8(e, SITE_FC1_DGRAD, w, 6, nullptr, 4 * C, M, C, 4 * C, a, EPI_BF16, s));
      } else {
      COATI_TRY(gemm(e, SITE_FC2_DGRAD, dxa, 0, C, e->S + w.fc2T, C, M, 4 * C, C, dh4, 4 * C, nullptr, EPI_MUL_AUX, p.hpre[l], nullptr, 4 * C, s));
      // hpre = a2 W1^T + b1 ; at packed-batch sizes with ln_2's backward in the write-out
      // ... and, chained behind it in the same launch, the c_proj input gradient dyb = dxb Wp (the attention backward's dO)
      ln2_fused = dgrad_lnbwd(SITE_FC1_DGRAD, dh4, 4 * C, e->S + w.fc1T, p.xmid[l], p.mean2[l], p.rstd2[l], e->P + w.ln2w, w.ln2w, w.ln2b, dxb,
                              e->S + w.projT, e->dyb);
      if (ln2_fused < 0) return ln2_fused;
      if (!ln2_fused) COATI_TRY(gemm(e, SITE_FC1_DGRAD, dh4, 0, 4 * C, e->S + w.fc1T, 4 * C, M, C, 4 * C, e->da, C, nullptr, EPI_BF16, nullptr, nullptr, 0, s));
      }
      if (!grp) {
        COATI_TRY(wgrad(e, SITE_XF_WGRAD, dh4, 0, 4 * C, p.a2[l], C, M, 4 * C, C, e->G + w.fc1w, C, e->G + w.fc1b, 0, s));
        // (the fc2 weight gradient runs after the two consumers of dh4, so that dh4 is re-read while it is still warm)
        COATI_TRY(wgrad(e, SITE_XF_WGRAD, dxa, 0, C, p.g[l], 4 * C, M, C, 4 * C, e->G + w.fc2w, 4 * C, e->G + w.fc2b, 0, s));
      }
      if (!ln2_fused) {
        ProfScope ps(e, SITE_LN_BWD, 0, s, (double)M * C * (2 + 4 + 4 + 4 + 2));   // dy16, x, dres in; dx, dx16 out
        COATI_TRY(ln_bwd(e->da, 0, p.xmid[l], p.mean2[l], p.rstd2[l], e->P + w.ln2w, DX, w.ln2w, w.ln2b, dxb));
      }
    }
    // xmid = x[l] + y Wp^T + bp
    if (c.use_fp8) {
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.C = e->dyb; a.ldc = C;
      COATI_TRY(gemm8(e, SITE_PROJ_DGRAD, w, 5, dxb, C, M, C, C, a, EPI_BF16, s));
    } else if (!ln2_fused) {   // (fused: the chained product of the launch above has written dyb)
    COATI_TRY(gemm(e, SITE_PROJ_DGRAD, dxb, 0, C, e->S + w.projT, C, M, C, C, e->dyb, C, nullptr, EPI_BF16, nullptr, nullptr, 0, s));
    }
    if (!grp) COATI_TRY(wgrad(e, SITE_XF_WGRAD, dxb, 0, C, p.y[l], C, M, C, C, e->G + w.projw, C, e->G + w.projb, 0, s));
    {
      ProfScope ps(e, SITE_ATTN_BWD, 10.0 * M * (double)p.T * C, s, (double)M * 8 * C * 2 + (double)M * c.n_head * 8);   // qkv, y, dy in; dqkv out
      COATI_TRY(launch_attn_bwd(p.qkv[l], p.y[l], e->dyb, p.lse[l], e->attnD, dqkv, e->cos_t, e->sin_t, p.B, p.T, c.n_head, C / c.n_head, s, p.packed ? p.off : nullptr, p.packed && lpt_on() ? p.ord : nullptr));
    }
    if (c.use_fp8) {
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.C = e->da; a.ldc = C;
      COATI_TRY(gemm8(e, SITE_QKV_DGRAD, w, 4, dqkv, 3 * C, M, C, 3 * C, a, EPI_BF16, s));
    } else {
    ln1_fused = dgrad_lnbwd(SITE_QKV_DGRAD, dqkv, 3 * C, e->S + w.attnT, p.x[l], p.mean1[l], p.rstd1[l], e->P + w.ln1w, w.ln1w, w.ln1b, dx_out);
    if (ln1_fused < 0) return ln1_fused;
    if (!ln1_fused) COATI_TRY(gemm(e, SITE_QKV_DGRAD, dqkv, 0, 3 * C, e->S + w.attnT, 3 * C, M, C, 3 * C, e->da, C, nullptr, EPI_BF16, nullptr, nullptr, 0, s));
    }
    if (!grp) COATI_TRY(wgrad(e, SITE_XF_WGRAD, dqkv, 0, 3 * C, p.a1[l], C, M, 3 * C, C, e->G + w.attnw, C, e->G + w.attnb, 0, s));
    if (!ln1_fused) {
      ProfScope ps(e, SITE_LN_BWD, 0, s, (double)M * C * (2 + 4 + 4 + 4 + 2));   // dy16, x, dres in; dx, dx16 out
      COATI_TRY(ln_bwd(e->da, 0, p.x[l], p.mean1[l], p.rstd1[l], e->P + w.ln1w, DX, w.ln1w, w.ln1b, dx_out));
    }
  }
  if (grp && l_hi > l_lo) COATI_TRY(xformer_wgrad_group(e, p, l_lo, l_hi, s));
  if (defer && fin.n > 0) {
    ProfScope ps(e, SITE_LN_BWD, 0, s, 0.0);
    COATI_TRY(launch_ln_finish_batched(e->ln_part_x, slot_stride, nblk, e->G, fin, C, s));
  }
  if (l_lo > 0) return COATI_OK;
  ProfScope ps(e, SITE_EMBED, 0, s);
  if (c.norm_embed) {
    // the injected rows' gradient goes to the injection and is CLEARED (the LayerNorm's output was overwritten there: nothing flows into
    // it), then the embedding LayerNorm's backward over all rows (its output lands in x[0], dead by now), then the table scatter
    if (dinjection != nullptr) {
      COATI_TRY(launch_embed_bwd(p.idx, DX, nullptr, dinjection, c.unk_token, p.B, p.T, C, c.n_tok, s, p.packed ? p.off : nullptr, 1));
      COATI_TRY(launch_embed_fwd(p.idx, nullptr, nullptr, c.unk_token, DX, p.B, p.T, C, c.n_tok, s, p.packed ? p.row_src : nullptr, M, 2));
    }
    COATI_TRY(launch_layernorm_bwd(DX, 1, C, p.x_emb, C, 0, p.mean0, p.rstd0, e->P + e->emb_lnw, nullptr, p.x[0], nullptr, e->G + e->emb_lnw, e->G + e->emb_lnb,
                                   e->ln_partial, M, C, s));
    return launch_embed_bwd(p.idx, p.x[0], e->G + e->tok_emb, nullptr, c.unk_token, p.B, p.T, C, c.n_tok, s, p.packed ? p.off : nullptr);
  }
  return launch_embed_bwd(p.idx, DX, e->G + e->tok_emb, dinjection, c.unk_token, p.B, p.T, C, c.n_tok, s, p.packed ? p.off : nullptr);
}

// ---- point encoder -------------------------------------------------------------------------------------
int gnn_fwd(coati_engine* e, const long long* atoms, const float* coords, hipStream_t s) {
  const coati_config& c = e->cfg;
  const int H = c.n_hidden_e3nn, Lg = c.n_layer_e3gnn, B = e->B, A = e->A, BA = B * A, Me = BA * A;
  {
    // embedding: atoms in, h (f32 + bf16) + rstd + mask out; geometry: coords in, dense d2 / w [B, A, A] out; compaction: the dense
    // grid in (twice: count, fill), the edge list (bj, bk, rev, d2, w: 20 B per edge) + offsets out
    ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (8 + H * 6 + 8) + (double)BA * 12 + (double)Me * 8 * 3 + (double)BA * 4, 20.0);
    bf16_t* h16 = Lg > 0 ? e->g_hcat[0] : e->g_hfin16;
    if (c.torch_emb) COATI_TRY(launch_gnn_embed(atoms, nullptr, nullptr, e->P + e->gembw, nullptr, e->g_h32[0], h16, Lg > 0 ? 2 * H : H, e->g_rstd[0], e->g_mask, BA, H, s, e->err_flag));
    else COATI_TRY(launch_gnn_embed(atoms, e->lut_ix, e->lut_iy, e->P + e->gembw, e->P + e->gembb, e->g_h32[0], h16, Lg > 0 ? 2 * H : H, e->g_rstd[0], e->g_mask, BA, H, s));
    COATI_TRY(launch_gnn_geom(coords, e->g_mask, c.msg_cutoff, e->g_d2, e->g_w, B, A, s));
    // neighbour list of the step (the coordinates do not change across the layers; the reference rebuilds it 5 times)
    COATI_TRY(launch_gnn_compact(e->g_w, e->g_d2, e->g_seg, e->g_ne, e->g_ebj, e->g_ebk, e->g_erev, e->g_ed2, e->g_ew, e->g_pos, B, A, s));
  }
  for (int l = 0; l < Lg; ++l) {
    const GLayerP& w = e->gl[l];
    COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_hcat[l], 0, 2 * H, e->S + w.w1ab, H, BA, 2 * H, H, e->g_P[l], 2 * H, nullptr, EPI_BF16, nullptr, nullptr, 0, s));
    if (H == 256 && gnn_edge_fused_on()) {
      // the three edge steps as ONE weight-resident launch (gemm_rb16.hip gnn_edge_fwd_fused_kernel): per edge the gathered sender row + bk + d2 + w
      // in, e1 and s2 out (once each, for the backward); per receiver its Pa row in, the segment sum out; the weight once
      ProfScope ps(e, SITE_GNN_EDGE_GEMM, 0, s, (double)BA * (H * 2 + 4 + H * 2) + (double)H * H * 2, (double)H * 2 + 12 + (double)H * 4, 2.0 * H * H + 6.0 * H);
      COATI_TRY(launch_gnn_edge_fwd_fused(e->g_P[l], 2 * H, e->g_seg, e->g_ebk, e->g_ed2, e->g_ew, e->P + w.e0w + 2 * H, 2 * H + 1, e->P + w.e0b,
                                          e->S + w.e3w, H, e->P + w.e3b, e->g_e1[l], e->g_s2[l], e->g_hcat[l] + H, 2 * H, BA, H, s));
    } else {
    {
      // per receiver its Pa row (bf16 H), per edge the GATHERED sender row Pb[bk] (bf16 H) + bk + d2 in, e1 (bf16 H) out
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 2 + 4), (double)H * 4 + 8, 4.0 * H);
      COATI_TRY(launch_gnn_edge_pre_c(e->g_P[l], 2 * H, e->g_seg, e->g_ebk, e->g_ed2, e->P + w.e0w + 2 * H, 2 * H + 1, e->P + w.e0b, e->g_e1[l], BA, H, s));
    }
    {   // rows = the edges that exist (device-side count); Me only sizes the grid
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.m_dev = e->g_ne;
      a.A = e->g_e1[l]; a.lda = H; a.B = e->S + w.e3w; a.ldb = H; a.M = Me; a.N = H; a.K = H; a.C = e->g_s2[l]; a.ldc = H; a.bias = e->P + w.e3b;
      ProfScope ps(e, SITE_GNN_EDGE_GEMM, 0, s, (double)H * H * 2 + H * 4, (double)H * 4, 2.0 * H * H);   // e1 in, s2 out per edge; the weight once
      COATI_TRY(launch_gemm_nt(a, 0, EPI_BF16, s));
    }
    {
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 2 + 4), (double)H * 2 + 4, 2.0 * H);   // s2 + w per edge in, the segment sums (bf16 H per node) out
      COATI_TRY(launch_gnn_edge_reduce_c(e->g_s2[l], e->g_seg, e->g_ew, e->g_hcat[l] + H, 2 * H, BA, H, s));
    }
    }
    if (c.residual) {
      // node_mlp(cat([h, mi, h0])) (e_gcl_sparse.py:282-290): the [h | mi] columns as a product (f32 out, into g_o: rewritten by the next
      // product), the one-hot h0 columns gathered on top, then the pre-activation / SiLU pair EPI_SILU writes otherwise
      COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_hcat[l], 0, 2 * H, e->S + w.n0p, 2 * H, BA, H, 2 * H, e->g_o, H, e->P + w.n0b, EPI_F32, nullptr, nullptr, 0, s));
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 4 + 8 + H * 4));
      COATI_TRY(launch_gnn_node_res_silu(e->g_o, atoms, e->lut_ix, e->lut_iy, e->P + w.n0w + 2 * H, 2 * H + 28, e->g_upre[l], e->g_t[l], BA, H, s));
    } else {
      COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_hcat[l], 0, 2 * H, e->S + w.n0w, 2 * H, BA, H, 2 * H, e->g_t[l], H, e->P + w.n0b, EPI_SILU, nullptr, e->g_upre[l], H, s));
    }
    COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_t[l], 0, H, e->S + w.n3w, H, BA, H, H, e->g_o, H, e->P + w.n3b, EPI_RES_F32, e->g_h32[l], nullptr, H, s));
    {
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 4 + H * 4 + H * 2 + 4));   // o in, h (f32) + its bf16 copy + rstd out
      const bool last = (l + 1 == Lg);
      COATI_TRY(launch_layernorm_fwd(e->g_o, H, nullptr, nullptr, last ? e->g_hfin16 : e->g_hcat[l + 1], last ? H : 2 * H, e->g_h32[l + 1], H, nullptr, e->g_rstd[l + 1], BA, H, s));
    }
  }
  COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_hfin16, 0, H, e->S + e->gd0w, H, BA, H, H, e->g_td, H, e->P + e->gd0b, EPI_SILU, nullptr, e->g_dpre, H, s));
  COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_td, 0, H, e->S + e->gd3w, H, BA, H, H, e->g_o2, H, e->P + e->gd3b, EPI_F32, nullptr, nullptr, 0, s));
  ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 4 + 4) + (double)B * H * 4);
  return launch_gnn_readout(e->g_o2, e->g_mask, e->hpoint, B, A, H, s);
}

// The node-level weight gradients of the point encoder (2 decoder Linears + 4 per layer, 16 384 rows each) as ONE launch at the end of
// gnn_bwd: a table of (problem, 128 x 128 tile, slice of M) entries, cached while the workspace layout stays the same
int gnn_wgrad_group(coati_engine* e, hipStream_t s) {
  const coati_config& c = e->cfg;
  const int H = c.n_hidden_e3nn, Lg = c.n_layer_e3gnn, BA = e->B * e->A;
  // (the split table slices M = B * A itself: the call's B and A belong to the key next to the capacity layout)
  const long long sig = (e->carve_sig * 1000003 + e->B) * 1000003 + e->A;
  if (e->gtab_sig != sig) {
    std::vector<WgradTile> tab;
    auto add = [&](const bf16_t* A_, int lda, const bf16_t* B_, int ldb, int N, int K, float* dW, int64_t ldw, float* db) -> int {
      WgradArgs a;
      a.A = A_; a.lda = lda; a.B = B_; a.ldb = ldb; a.M = BA; a.N = N; a.K = K; a.dW = dW; a.ldw = ldw; a.dbias = db; a.n_out = 0;
      return wgrad_table_append_split(tab, a, 4);
    };
    COATI_TRY(add(e->g_do2, H, e->g_td, H, H, H, e->G + e->gd3w, H, e->G + e->gd3b));
    COATI_TRY(add(e->g_dtd, H, e->g_hfin16, H, H, H, e->G + e->gd0w, H, e->G + e->gd0b));
    for (int l = Lg - 1; l >= 0; --l) {
      const GLayerP& w = e->gl[l];
      COATI_TRY(add(e->gl_DO16[l], H, e->g_t[l], H, H, H, e->G + w.n3w, H, e->G + w.n3b));
      COATI_TRY(add(e->gl_du[l], H, e->g_hcat[l], 2 * H, H, 2 * H, e->G + w.n0w, 2 * H + (c.residual ? 28 : 0), e->G + w.n0b));
      COATI_TRY(add(e->gl_dP[l], 2 * H, e->g_hcat[l], 2 * H, H, H, e->G + w.e0w, 2 * H + 1, nullptr));
      COATI_TRY(add(e->gl_dP[l] + H, 2 * H, e->g_hcat[l], 2 * H, H, H, e->G + w.e0w + H, 2 * H + 1, nullptr));
    }
    COATI_CHECK_ARG((int)tab.size() <= e->gtab_cap, "gnn wgrad group: table overflow (%zu > %d)", tab.size(), e->gtab_cap);
    COATI_TRY(upload_table(e, 4 + (e->gtab_rr++ & 1), tab, e->d_gtab, s, "gnn wgrad group"));
    e->gtab_n = (int)tab.size();
    e->gtab_sig = sig;
  }
  const double nprob = 2.0 + 4.0 * Lg + Lg;   // (the H x 2H product counts twice)
  ProfScope ps(e, SITE_GNN_WGRAD, 2.0 * BA * H * H * nprob, s, nprob * ((double)BA * H * 4 + (double)H * H * 8));
  return launch_wgrad_split_table(e->d_gtab, e->gtab_n, s);
}

int gnn_bwd(coati_engine* e, const float* dhpoint, hipStream_t s) {
  const coati_config& c = e->cfg;
  const int H = c.n_hidden_e3nn, Lg = c.n_layer_e3gnn, B = e->B, A = e->A, BA = B * A, Me = BA * A;
  float *DH = e->g_DH, *DO = e->g_DO;
  const bool grp = e->gnn_wg_group;
  {
    ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)B * H * 4 + (double)BA * (H * 2 + 4));
    COATI_TRY(launch_gnn_readout_bwd(dhpoint, e->g_mask, e->g_do2, B, A, H, s));
  }
  COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_do2, 0, H, e->S + e->gd3T, H, BA, H, H, e->g_dtd, H, nullptr, EPI_DSILU, e->g_dpre, nullptr, H, s));
  if (!grp) COATI_TRY(wgrad(e, SITE_GNN_WGRAD, e->g_do2, 0, H, e->g_td, H, BA, H, H, e->G + e->gd3w, H, e->G + e->gd3b, 0, s));
  COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, e->g_dtd, 0, H, e->S + e->gd0T, H, BA, H, H, DH, H, nullptr, EPI_F32, nullptr, nullptr, 0, s));
  if (!grp) COATI_TRY(wgrad(e, SITE_GNN_WGRAD, e->g_dtd, 0, H, e->g_hfin16, H, BA, H, H, e->G + e->gd0w, H, e->G + e->gd0b, 0, s));
  for (int l = Lg - 1; l >= 0; --l) {
    const GLayerP& w = e->gl[l];
    // grouped weight gradients: this layer's gradient operands stay alive in their own buffers until the launch at the end
    bf16_t* const DO16 = grp ? e->gl_DO16[l] : e->g_DO16;
    bf16_t* const du = grp ? e->gl_du[l] : e->g_du;
    bf16_t* const dP = grp ? e->gl_dP[l] : e->g_dP;
    {
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 4 + H * 4 + 4 + H * 4 + H * 2));   // dy, xhat, rstd in; dx (f32 + bf16) out
      COATI_TRY(launch_layernorm_bwd(DH, 1, H, e->g_h32[l + 1], H, 1, nullptr, e->g_rstd[l + 1], nullptr, nullptr, DO, DO16, nullptr, nullptr, e->ln_partial, BA, H, s));
    }
    // o = h + t W4^T + b4
    COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, DO16, 0, H, e->S + w.n3T, H, BA, H, H, du, H, nullptr, EPI_DSILU, e->g_upre[l], nullptr, H, s));
    if (!grp) COATI_TRY(wgrad(e, SITE_GNN_WGRAD, DO16, 0, H, e->g_t[l], H, BA, H, H, e->G + w.n3w, H, e->G + w.n3b, 0, s));
    // u = [h | mi] W3^T + b3 ; W3T is [2H rows][H]
    COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, du, 0, H, e->S + w.n0T, H, BA, H, H, DO, H, nullptr, EPI_ACC_F32, nullptr, nullptr, 0, s));
    COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, du, 0, H, e->S + w.n0T + (int64_t)H * H, H, BA, H, H, e->g_dmi, H, nullptr, EPI_BF16, nullptr, nullptr, 0, s));
    if (!grp) COATI_TRY(wgrad(e, SITE_GNN_WGRAD, du, 0, H, e->g_hcat[l], 2 * H, BA, H, 2 * H, e->G + w.n0w, 2 * H + (c.residual ? 28 : 0), e->G + w.n0b, 0, s));
    if (c.residual) {   // the h0 columns of node_mlp.0.weight: one-hot scatter of du
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 2 + 8));
      COATI_TRY(launch_gnn_onehot_wgrad(e->atoms, e->lut_ix, e->lut_iy, du, e->G + w.n0w + 2 * H, 2 * H + 28, BA, H, s));
    }
    {
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 2 + 4), (double)H * 4 + 4, 6.0 * H);   // d mi per receiver, s2 + w per edge in, d s2 per edge out
      COATI_TRY(launch_gnn_edge_reduce_bwd_c(e->g_dmi, H, e->g_s2[l], e->g_seg, e->g_ew, e->g_ds2, BA, H, s));
    }
    {
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.A = e->g_ds2; a.lda = H; a.B = e->S + w.e3T; a.ldb = H; a.M = Me; a.N = H; a.K = H; a.C = e->g_dpre1; a.ldc = H;
      a.P = e->g_P[l]; a.ldp = 2 * H; a.d2 = e->g_ed2; a.w1c = e->P + w.e0w + 2 * H; a.w1c_stride = 2 * H + 1;
      a.b1 = e->P + w.e0b; a.natom = A; a.H = H;
      a.m_dev = e->g_ne; a.e_bj = e->g_ebj; a.e_bk = e->g_ebk;
      // d s2 in, d pre out, and the recomputed pre-activation's two GATHERED rows Pa[bj], Pb[bk] (bf16 H each) + the indices and d2 per edge
      ProfScope ps(e, SITE_GNN_EDGE_GEMM, 0, s, (double)H * H * 2, (double)H * 8 + 12, 2.0 * H * H + 8.0 * H);
      COATI_TRY(launch_gemm_nt(a, 0, EPI_EDGE_DPRE, s));
    }
    {
      WgradArgs a;
      a.A = e->g_ds2; a.lda = H; a.B = e->g_e1[l]; a.ldb = H; a.M = Me; a.N = H; a.K = H; a.dW = e->G + w.e3w; a.ldw = H; a.dbias = e->G + w.e3b;
      a.n_out = 0; a.m_dev = e->g_ne;
      ProfScope ps(e, SITE_GNN_WGRAD, 0, s, (double)H * H * 8, (double)H * 4, 2.0 * H * H);   // d s2 and e1 per edge in, the f32 gradient read-modify-write
      COATI_TRY(launch_wgrad(a, 0, s));
    }
    {
      // d pre of an edge in (its own row for the receiver sum, the reverse edge's row GATHERED for the sender sum) + rev + d2; dP (bf16 2H per node) out
      ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 4 + 4), (double)H * 4 + 8, 4.0 * H);
      COATI_TRY(launch_gnn_edge_pre_bwd_c(e->g_dpre1, e->g_seg, e->g_erev, e->g_ed2, dP, 2 * H, e->G + w.e0w + 2 * H, 2 * H + 1, e->G + w.e0b, BA, H, s));
    }
    COATI_TRY(gemm(e, SITE_GNN_NODE_GEMM, dP, 0, 2 * H, e->S + w.w1abT, 2 * H, BA, H, 2 * H, DO, H, nullptr, EPI_ACC_F32, nullptr, nullptr, 0, s));
    if (!grp) {
      COATI_TRY(wgrad(e, SITE_GNN_WGRAD, dP, 0, 2 * H, e->g_hcat[l], 2 * H, BA, H, H, e->G + w.e0w, 2 * H + 1, nullptr, 0, s));
      COATI_TRY(wgrad(e, SITE_GNN_WGRAD, dP + H, 0, 2 * H, e->g_hcat[l], 2 * H, BA, H, H, e->G + w.e0w + H, 2 * H + 1, nullptr, 0, s));
    }
    float* t = DH; DH = DO; DO = t;
  }
  if (grp) COATI_TRY(gnn_wgrad_group(e, s));
  {
    ProfScope ps(e, SITE_GNN_ELEMWISE, 0, s, (double)BA * (H * 4 + H * 4 + 4 + H * 4) + (double)BA * (8 + H * 4) + 30.0 * H * 4);
    COATI_TRY(launch_layernorm_bwd(DH, 1, H, e->g_h32[0], H, 1, nullptr, e->g_rstd[0], nullptr, nullptr, DO, nullptr, nullptr, nullptr, e->ln_partial, BA, H, s));
    if (e->cfg.torch_emb) COATI_TRY(launch_gnn_embed_bwd(e->atoms, nullptr, nullptr, DO, e->G + e->gembw, nullptr, BA, H, s));
    else COATI_TRY(launch_gnn_embed_bwd(e->atoms, e->lut_ix, e->lut_iy, DO, e->G + e->gembw, e->G + e->gembb, BA, H, s));
  }
  return COATI_OK;
}

// fp32 Linear backward on [B, *] head tensors: dW += dY^T X ; db += colsum(dY) ; dX = dY W
// (the three products join a batch: independent heads share one launch, gemm.hip sgemm_batch_kernel)
int head_linear_bwd(coati_engine* e, SgemmBatch& sb, const float* dY, const float* X, int64_t w_off, int64_t b_off, float* dX, int B,
                    int N, int K) {
  const float* W = e->P + w_off;
  COATI_TRY(sgemm_batch_add(sb, dY, 1, N, X, K, 1, e->G + w_off, K, N, K, B, nullptr, 1.f, 1));          // dW[n,k] += sum_b dY[b,n] X[b,k]
  COATI_TRY(sgemm_batch_add(sb, e->ones, 0, 1, dY, N, 1, e->G + b_off, N, 1, N, B, nullptr, 1.f, 1));    // db[n] += sum_b dY[b,n]
  if (dX) COATI_TRY(sgemm_batch_add(sb, dY, N, 1, W, K, 1, dX, K, B, K, N, nullptr, 1.f, 0));            // dX[b,k] = sum_n dY[b,n] W[n,k]
  return COATI_OK;
}

int ensure_side(coati_engine* e) {
  if (e->side) return COATI_OK;
  // (a high-priority side stream was tried in round 2: no change -- 34.8 ms either way; with COATI_NO_OVERLAP=1 the step
  // takes 34.85 ms: the transformer's kernels hold every CU's LDS, so the point encoder's kernels run in their tails)
  if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) {
    coati_set_error("engine: could not create the side stream");
    return COATI_EHIP;
  }
  return COATI_OK;
}
// side waits for everything enqueued on main so far
int fork_side(coati_engine* e, hipStream_t main) {
  COATI_TRY(ensure_side(e));
  if (hipEventRecord(e->ev_fork, main) != hipSuccess || hipStreamWaitEvent(e->side, e->ev_fork, 0) != hipSuccess) {
    coati_set_error("engine: fork failed");
    return COATI_EHIP;
  }
  return COATI_OK;
}
// main waits for everything enqueued on side so far
int join_side(coati_engine* e, hipStream_t main) {
  if (hipEventRecord(e->ev_join, e->side) != hipSuccess || hipStreamWaitEvent(main, e->ev_join, 0) != hipSuccess) {
    coati_set_error("engine: join failed");
    return COATI_EHIP;
  }
  return COATI_OK;
}

// point_to_clip / smiles_to_clip (clip_e2e.py:405-428): LayerNorm -> Linear (norm_clips) or a plain Linear, f32 on [B, *] rows
int point_head_fwd(coati_engine* e, float* out, hipStream_t s) {
  const coati_config& c = e->cfg;
  const int H = c.n_hidden_e3nn, E = c.n_embd_common, B = e->B;
  if (!c.use_point_encoder) return COATI_OK;   // zeros, written by the caller
  const float* x = e->hpoint;
  if (c.norm_clips && c.old_architecture) {
    // Linear -> LayerNorm (clip_e2e.py:409-413): the Linear's output is kept in hp_ln ([B, H] = [B, E]) for the LayerNorm's backward
    COATI_TRY(launch_sgemm(x, H, 1, e->P + e->p2c_w, 1, H, e->hp_ln, E, B, E, H, e->P + e->p2c_b, 1.f, 0, s));
    return launch_layernorm_fwd(e->hp_ln, E, e->P + e->p2c_lnw, e->P + e->p2c_lnb, nullptr, 0, out, E, e->hp_mean, e->hp_rstd, B, E, s);
  }
  if (c.norm_clips) {
    COATI_TRY(launch_layernorm_fwd(e->hpoint, H, e->P + e->p2c_lnw, e->P + e->p2c_lnb, nullptr, 0, e->hp_ln, H, e->hp_mean, e->hp_rstd, B, H, s));
    x = e->hp_ln;
  }
  return launch_sgemm(x, H, 1, e->P + e->p2c_w, 1, H, out, E, B, E, H, e->P + e->p2c_b, 1.f, 0, s);
}
int smiles_head_fwd(coati_engine* e, float* out, hipStream_t s) {
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer, E = c.n_embd_common, B = e->B;
  const float* x = e->hstop;
  if (c.norm_clips && c.old_architecture) {   // Linear -> LayerNorm (clip_e2e.py:414-417); E == C
    COATI_TRY(launch_sgemm(x, C, 1, e->P + e->s2c_w, 1, C, e->hs_ln, E, B, E, C, e->P + e->s2c_b, 1.f, 0, s));
    return launch_layernorm_fwd(e->hs_ln, E, e->P + e->s2c_lnw, e->P + e->s2c_lnb, nullptr, 0, out, E, e->hs_mean, e->hs_rstd, B, E, s);
  }
  if (c.norm_clips) {
    COATI_TRY(launch_layernorm_fwd(e->hstop, C, e->P + e->s2c_lnw, e->P + e->s2c_lnb, nullptr, 0, e->hs_ln, C, e->hs_mean, e->hs_rstd, B, C, s));
    x = e->hs_ln;
  }
  return launch_sgemm(x, C, 1, e->P + e->s2c_w, 1, C, out, E, B, E, C, e->P + e->s2c_b, 1.f, 0, s);
}

}  // namespace

// =====================================================================================================
// C ABI: engine
// =====================================================================================================
extern "C" {

int coati_engine_create(const coati_config* cfg, coati_engine** out) {
  COATI_CHECK_ARG(cfg && out, "engine_create: null argument");
  const int C = cfg->n_hidden_xformer, H = cfg->n_hidden_e3nn, E = cfg->n_embd_common;
  COATI_CHECK_SHAPE(cfg->n_head > 0 && (C == cfg->n_head * 16 || C == cfg->n_head * 32), "engine_create: head size must be 16 or 32 (C=%d, n_head=%d)", C, cfg->n_head);
  COATI_CHECK_SHAPE(C % 64 == 0 && H % 64 == 0 && C <= 1024 && H <= 1024, "engine_create: C=%d and H=%d must be multiples of 64 (<=1024)", C, H);
  COATI_CHECK_SHAPE(E == C, "engine_create: n_embd_common (%d) must equal n_hidden_xformer (%d)", E, C);
  COATI_CHECK_SHAPE(cfg->n_seq > 0 && cfg->n_seq <= 256 && cfg->n_tok > 8, "engine_create: n_seq must be <= 256");
  COATI_CHECK_SHAPE(cfg->n_layer_xformer >= 1 && cfg->n_layer_e3gnn >= 0, "engine_create: bad layer counts");
  COATI_CHECK_SHAPE(!cfg->use_fp8 || C % 128 == 0, "engine_create: fp8 mode needs n_hidden_xformer %% 128 == 0 (C=%d)", C);
  COATI_CHECK_ARG(!(cfg->residual && cfg->torch_emb), "engine_create: residual and torch_emb together size the node MLPs for 28 one-hot features and feed them H embedding columns (the reference fails in its first forward: e3gnn_clip.py:100, e_gcl_sparse.py:289)");
  COATI_CHECK_SHAPE(!(cfg->old_architecture && cfg->norm_clips) || H == E, "engine_create: old_architecture needs n_hidden_e3nn == n_embd_common (%d, %d): point_to_clip's LayerNorm is sized by the one and applied to the other (clip_e2e.py:410-413)", H, E);
  coati_engine* e = new coati_engine();
  e->cfg = *cfg;
  build_layout(e);
  *out = e;
  return COATI_OK;
}

void coati_engine_destroy(coati_engine* e) {
  if (!e) return;
  for (auto& g : e->dec.graph)
    if (g) hipGraphExecDestroy(g);
  for (auto ev : e->ev) hipEventDestroy(ev);
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  if (e->ev_join) hipEventDestroy(e->ev_join);
  if (e->side) hipStreamDestroy(e->side);
  for (int i = 0; i < 6; ++i) {
    if (e->h_tab_ev[i]) hipEventDestroy(e->h_tab_ev[i]);
    if (e->h_tab[i]) hipHostFree(e->h_tab[i]);
  }
  delete e;
}

int64_t coati_engine_param_elems(const coati_engine* e) { return e ? e->n_params : 0; }
int64_t coati_engine_trainable_elems(const coati_engine* e) { return e ? e->n_trainable : 0; }
int coati_engine_n_entries(const coati_engine* e) { return e ? (int)e->entries.size() : 0; }
int coati_engine_entry(const coati_engine* e, int i, char* name, int name_cap, int64_t* offset, int32_t* rows, int32_t* cols) {
  COATI_CHECK_ARG(e && i >= 0 && i < (int)e->entries.size() && name && name_cap > 0, "engine_entry: bad argument");
  const Entry& en = e->entries[i];
  snprintf(name, name_cap, "%s", en.name.c_str());
  if (offset) *offset = en.off;
  if (rows) *rows = en.rows;
  if (cols) *cols = en.cols;
  return COATI_OK;
}
int64_t coati_engine_shadow_elems(const coati_engine* e) { return e ? e->n_shadow : 0; }

// Grow-only capacities: every later carve (workspace_bytes / forward / encode) sizes its buffers for at least this shape, so that
// batches of different T / A keep the same buffer addresses (see coati_engine::cap_B).  Values below the current capacity are ignored.
int coati_engine_reserve(coati_engine* e, int B, int T1, int T2, int A) {
  COATI_CHECK_ARG(e && B >= 0 && T1 >= 0 && T2 >= 0 && A >= 0, "engine_reserve: bad argument");
  COATI_CHECK_SHAPE(T1 <= e->cfg.n_seq && T2 <= e->cfg.n_seq, "engine_reserve: T1=%d T2=%d beyond n_seq=%d", T1, T2, e->cfg.n_seq);
  if (B > e->cap_B) e->cap_B = B;
  if (T1 > e->cap_T1) e->cap_T1 = T1;
  if (T2 > e->cap_T2) e->cap_T2 = T2;
  if (A > e->cap_A) e->cap_A = A;
  return COATI_OK;
}

int64_t coati_engine_workspace_bytes(const coati_engine* e, int B, int T1, int T2, int A, int Bg) {
  if (!e || B <= 0 || T1 <= 0 || T2 <= 0 || A <= 0) return 0;
  coati_engine tmp = *e;  // carve on a copy: no side effects on the live engine
  tmp.ev.clear();
  Arena ar{nullptr, 0, 0, true};
  return (int64_t)carve(&tmp, ar, B, T1, T2, A, Bg);
}

int coati_engine_bind(coati_engine* e, float* params, float* grads, float* adam_m, float* adam_v, uint16_t* shadow,
                      const float* rope_cos, const float* rope_sin, const int32_t* lut_ix, const int32_t* lut_iy) {
  COATI_CHECK_ARG(e && params && shadow && rope_cos && rope_sin && lut_ix && lut_iy, "engine_bind: null argument");
  e->P = params; e->G = grads; e->Mo = adam_m; e->Vo = adam_v; e->S = shadow;
  e->cos_t = rope_cos; e->sin_t = rope_sin; e->lut_ix = lut_ix; e->lut_iy = lut_iy;
  return COATI_OK;
}

int64_t coati_engine_fp8_bytes(const coati_engine* e) { return e ? e->n_fp8 : 0; }
int coati_engine_bind_fp8(coati_engine* e, uint8_t* fp8_shadow, int64_t bytes) {
  COATI_CHECK_ARG(e && e->cfg.use_fp8 && fp8_shadow && bytes >= e->n_fp8, "engine_bind_fp8: not an fp8 engine / buffer too small");
  e->S8 = fp8_shadow;
  return COATI_OK;
}

// fp8 mode: MXFP8 copies of the transformer weights from the (fresh) bf16 shadows: natural [N, K] for the forward products,
// the transposed shadows for the input-gradient products (each quantised along ITS contraction dimension)
static int refresh_fp8_weights(coati_engine* e, hipStream_t s) {
  if (!e->cfg.use_fp8) return COATI_OK;
  COATI_CHECK_ARG(e->S8, "fp8 mode: coati_engine_bind_fp8 has not been called");
  const int C = e->cfg.n_hidden_xformer;
  for (const XLayerP& w : e->xl) {
    const int64_t src[8] = {w.attnw, w.projw, w.fc1w, w.fc2w, w.attnT, w.projT, w.fc1T, w.fc2T};
    const int rows[8] = {3 * C, C, 4 * C, C, C, C, C, 4 * C}, cols[8] = {C, C, C, 4 * C, 3 * C, C, 4 * C, C};
    for (int i = 0; i < 8; ++i)
      COATI_TRY(launch_quant_mx8(e->S + src[i], 0, cols[i], e->S8 + w.q8[i], cols[i], e->S8 + w.s8[i], rows[i], cols[i], s));
  }
  return COATI_OK;
}

static int refresh_shadows_impl(coati_engine* e, void* stream, bool natural_done) {
  COATI_CHECK_ARG(e && e->P && e->S, "refresh_shadows: engine not bound");
  hipStream_t s = (hipStream_t)stream;
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer, H = c.n_hidden_e3nn;
  ProfScope ps(e, SITE_OPTIM, 0, s);
  if (!natural_done) COATI_TRY(launch_cast_bf16(e->P, e->S, e->n_params, s));
  if (e->d_jobs != nullptr) {
    if (!e->jobs_uploaded) {
      if (hipMemcpyAsync(e->d_jobs, e->jobs.data(), e->jobs.size() * sizeof(ShadowJob), hipMemcpyHostToDevice, s) != hipSuccess ||
          hipMemcpyAsync(e->d_tile_start, e->tile_start.data(), e->tile_start.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) {
        coati_set_error("refresh_shadows: job table upload failed");
        return COATI_EHIP;
      }
      e->jobs_uploaded = true;
    }
    COATI_TRY(launch_shadow_jobs(e->d_jobs, e->d_tile_start, (int)e->jobs.size(), e->n_job_tiles, e->P, e->S, s));
    return refresh_fp8_weights(e, s);
  }
  // no workspace yet (first refresh after loading weights): one launch per matrix
  for (const ShadowJob& j : e->jobs) {
    if (j.transpose) COATI_TRY(launch_transpose_cast(e->P + j.src_off, j.ld_src, e->S + j.dst_off, j.ld_dst, j.rows, j.cols, s));
    else COATI_TRY(launch_pack_rows_cast(e->P + j.src_off, j.ld_src, e->S + j.dst_off, j.ld_dst, j.rows, j.cols, s));
  }
  return refresh_fp8_weights(e, s);
}

int coati_engine_refresh_shadows(coati_engine* e, void* stream) { return refresh_shadows_impl(e, stream, false); }

// decoder pass with injection + lm_head / AR cross-entropy: the second half of coati_engine_forward
static int forward_decoder_impl(coati_engine* e, hipStream_t s) {
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer;
  float* const scal = e->scal;
  // ---- decoder pass with injection (smiles_xformer.py:426-452) ----
  COATI_TRY(xformer_fwd(e, e->p2, e->cliptok, s));
  // ---- lm_head + AR cross-entropy, logits never materialised (smiles_xformer.py:453, train_coati.py:260-265) ----
  if (e->y_next) {
    const int M2 = e->p2.M;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = e->p2.af; a.lda = C; a.B = e->S + e->lmhead; a.ldb = C; a.M = M2; a.N = c.n_tok; a.K = C; a.partial = e->ce_partial;
    a.partial_tile = 64;   // ce_partial is sized for 64-column entries: the row-block kernel may take the product
    GemmArgs aw = a;             // (the width is a property of the kernel that runs: with a row split, of a launch's row range)
    if (const int rows = row_split_plan(a, 0, EPI_CE_PARTIAL)) aw.M = rows;
    const int tiles_v = cdiv(c.n_tok, gemm_ce_tile_width(aw));
    {
      ProfScope ps(e, SITE_LMHEAD_FWD, 2.0 * M2 * c.n_tok * C, s, (double)M2 * C * 2 + (double)c.n_tok * C * 2);   // logits never leave the chip
      COATI_TRY(gemm_rows(a, 0, EPI_CE_PARTIAL, s));
    }
    COATI_TRY(launch_ce_finish(e->ce_partial, tiles_v, e->p2.af, C, e->S + e->lmhead, C, e->p2.packed ? e->p2.ypk : e->y_next, e->ce_lse, scal, M2, C, c.n_tok, s));
  }
  if (hipMemcpyAsync(scal + 6, e->err_flag, sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    coati_set_error("engine_forward: error-word copy failed");
    return COATI_EHIP;
  }
  e->have_fwd = true;
  return COATI_OK;
}

int coati_engine_forward(coati_engine* e, void* workspace, int64_t workspace_bytes, int B, int T1, int T2, int A,
                         const int64_t* raw_tokens, const int64_t* tokens, const int64_t* y_next,
                         const int64_t* atoms, const float* coords, const uint8_t* use_point, float* h_e3gnn,
                         float* h_smiles, uint8_t* bad_rows, float* scal, int train, int64_t rows1, int64_t rows2, void* stream) {
  COATI_CHECK_ARG(e && e->P && e->S, "engine_forward: engine not bound");
  COATI_CHECK_ARG(workspace && raw_tokens && tokens && atoms && coords && use_point && scal, "engine_forward: null argument");
  const bool stop_after_heads = (train & 2) != 0;   // train | 2: return behind the heads, coati_engine_forward_decoder runs the rest
  train &= 1;
  COATI_CHECK_ARG(!train || (e->G && y_next), "engine_forward: training needs grads and y_next");
  const coati_config& c = e->cfg;
  COATI_CHECK_SHAPE(B > 0 && T1 > 0 && T2 > 0 && A > 0 && T1 <= c.n_seq && T2 <= c.n_seq,
                    "engine_forward: unsupported shape B=%d T1=%d T2=%d A=%d (n_seq=%d)", B, T1, T2, A, c.n_seq);
  COATI_CHECK_SHAPE(rows1 >= 0 && rows2 >= 0 && rows1 <= (int64_t)B * T1 && rows2 <= (int64_t)B * T2 && (rows1 > 0) == (rows2 > 0),
                    "engine_forward: packed row counts %lld / %lld do not fit %d x %d / %d x %d", (long long)rows1, (long long)rows2, B, T1, B, T2);
  hipStream_t s = (hipStream_t)stream;
  const int C = c.n_hidden_xformer, H = c.n_hidden_e3nn, E = c.n_embd_common;
  Arena ar{reinterpret_cast<char*>(workspace), 0, (size_t)workspace_bytes, false};
  const size_t need = carve(e, ar, B, T1, T2, A, B);
  COATI_CHECK_ARG((int64_t)need <= workspace_bytes, "engine_forward: workspace too small (%zu > %lld)", need, (long long)workspace_bytes);
  if (e->nce) e->nce_cap = ((size_t)workspace_bytes - (size_t)(reinterpret_cast<char*>(e->nce) - reinterpret_cast<char*>(workspace))) / sizeof(float);
  e->B = B; e->T1 = T1; e->T2 = T2; e->A = A;
  e->p1.idx = reinterpret_cast<const long long*>(raw_tokens);
  e->p2.idx = reinterpret_cast<const long long*>(tokens);
  e->y_next = reinterpret_cast<const long long*>(y_next);
  e->atoms = reinterpret_cast<const long long*>(atoms);
  e->use_point = use_point;
  e->scal = scal;
  e->have_fwd = false;
  e->decoder_pending = false;

#define HIPCHK(x) do { hipError_t _h = (x); if (_h != hipSuccess) { coati_set_error("%s: %s", #x, hipGetErrorString(_h)); return COATI_EHIP; } } while (0)
  HIPCHK(hipMemsetAsync(scal, 0, 16 * sizeof(float), s));
  HIPCHK(hipMemsetAsync(e->err_flag, 0, 4 * sizeof(int), s));
  if (train) HIPCHK(hipMemsetAsync(e->G, 0, (size_t)e->n_params * sizeof(float), s));
  // ---- packed rows: both transformer passes run on the rows' real prefixes only (embed.hip, launch_seq_pack); the counts
  // come from the caller (the batch assembler knows them on the host: no device -> host sync here), the device checks them
  if (rows1 > 0) {
    COATI_TRY(launch_seq_pack(e->p1.idx, nullptr, c.pad_token, B, T1, (int)rows1, e->p1.off, e->p1.row_src, e->p1.row_t, nullptr, e->err_flag, s, e->p1.ord));
    COATI_TRY(launch_seq_pack(e->p2.idx, e->y_next, c.pad_token, B, T2, (int)rows2, e->p2.off, e->p2.row_src, e->p2.row_t, e->y_next ? e->p2.ypk : nullptr, e->err_flag, s, e->p2.ord));
    e->p1.packed = e->p2.packed = true;
    e->p1.M = (int)rows1;
    e->p2.M = (int)rows2;
  }
  {
    // ones[B] for bias column sums
    std::vector<float> dummy;  // (filled on device below)
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->ones, 0x3f800000, B, s));
  }

  // ---- point encoder (clip_e2e.py:454-461): on the side stream, concurrent with the encoder pass ----
  const bool ovl = e->overlap && (e->prof_mask == 0 || e->prof_keep_overlap) && c.use_point_encoder;
  if (!c.use_point_encoder) {
    // encode_points returns zeros (clip_e2e.py:462-463)
    HIPCHK(hipMemsetAsync(e->h_e3gnn, 0, (size_t)B * E * sizeof(float), s));
  } else if (ovl) {
    COATI_TRY(fork_side(e, s));
    COATI_TRY(gnn_fwd(e, e->atoms, coords, e->side));
  } else {
    COATI_TRY(gnn_fwd(e, e->atoms, coords, s));
  }
  // ---- encoder pass (clip_e2e.py:448-452) ----
  // (the [STOP] positions first: a training step runs the tail of the last layer on those rows only, XPass::tail)
  COATI_TRY(launch_find_stop(e->p1.idx, c.stop_token, e->stop_pos, e->err_flag, B, T1, s));
  {
    static const bool no_tail = getenv("COATI_NO_TAIL") != nullptr;   // A/B switch: every row through the whole last layer and ln_f
    e->p1.tail = train && !no_tail && !c.use_fp8;
    e->p2.tail = false;
  }
  COATI_TRY(xformer_fwd(e, e->p1, nullptr, s));
  if (ovl) COATI_TRY(join_side(e, s));
  COATI_TRY(point_head_fwd(e, e->h_e3gnn, s));
  // ---- smiles_to_clip ----
  if (e->p1.tail) HIPCHK(hipMemcpyAsync(e->hstop, e->p1.t_xf, (size_t)B * C * sizeof(float), hipMemcpyDeviceToDevice, s));
  else COATI_TRY(launch_gather_rows(e->p1.xf32, e->stop_pos, e->hstop, B, T1, C, s, e->p1.packed ? e->p1.off : nullptr));
  COATI_TRY(smiles_head_fwd(e, e->h_smiles, s));
  // ---- special token (clip_e2e.py:800-808): SiLU -> Linear of either embedding, or the embeddings themselves (token_mlp = False:
  // nn.Identity, clip_e2e.py:436-437) ----
  const float *ptok = e->h_e3gnn, *stok = e->h_smiles;
  if (c.token_mlp) {
    COATI_TRY(launch_silu_fwd(e->h_e3gnn, e->sa, (long long)B * E, s));
    COATI_TRY(launch_silu_fwd(e->h_smiles, e->sb, (long long)B * E, s));
    SgemmBatch sb;
    COATI_TRY(sgemm_batch_add(sb, e->sa, E, 1, e->P + e->tokw, 1, E, e->ptok, E, B, E, E, e->P + e->tokb, 1.f, 0));
    COATI_TRY(sgemm_batch_add(sb, e->sb, E, 1, e->P + e->tokw, 1, E, e->stok, E, B, E, E, e->P + e->tokb, 1.f, 0));
    COATI_TRY(launch_sgemm_batch(sb, s));
    ptok = e->ptok; stok = e->stok;
  }
  COATI_TRY(launch_select_rows(use_point, ptok, stok, e->cliptok, B, E, s));
  if (bad_rows) COATI_TRY(launch_bad_rows(e->p2.idx, bad_rows, B, T2, s));
  if (h_e3gnn) HIPCHK(hipMemcpyAsync(h_e3gnn, e->h_e3gnn, (size_t)B * E * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (h_smiles) HIPCHK(hipMemcpyAsync(h_smiles, e->h_smiles, (size_t)B * E * sizeof(float), hipMemcpyDeviceToDevice, s));
  e->have_ws = true;
  if (stop_after_heads) {
    // the embeddings are out: the caller may run the contrastive head (+ its collectives) on another stream while
    // coati_engine_forward_decoder enqueues the decoder pass on this one
    e->decoder_pending = true;
    return COATI_OK;
  }
  return forward_decoder_impl(e, s);
}

int coati_engine_forward_decoder(coati_engine* e, void* stream) {
  COATI_CHECK_ARG(e && e->decoder_pending, "engine_forward_decoder: no forward stopped behind the heads (train | 2)");
  e->decoder_pending = false;
  return forward_decoder_impl(e, (hipStream_t)stream);
}

// Inference encoders alone (clip_e2e.py:448-452 encode_tokens, :454-463 encode_points): only the requested tower runs --
// the encoder pass + [STOP] pick + smiles_to_clip when raw_tokens is given, the point encoder + point_to_clip when
// atoms / coords are given.  No decoder pass, no lm_head, nothing saved for a backward (have_fwd stays false).
int coati_engine_encode(coati_engine* e, void* workspace, int64_t workspace_bytes, int B, int T1, int A,
                        const int64_t* raw_tokens, const int64_t* atoms, const float* coords, float* h_smiles,
                        float* h_e3gnn, float* scal, void* stream) {
  COATI_CHECK_ARG(e && e->P && e->S && workspace && scal, "engine_encode: engine not bound / null argument");
  COATI_CHECK_ARG((raw_tokens && h_smiles) || (atoms && coords && h_e3gnn), "engine_encode: nothing to encode");
  const coati_config& c = e->cfg;
  const bool do_tok = raw_tokens && h_smiles, do_pts = atoms && coords && h_e3gnn;
  if (!do_tok) T1 = 1;
  if (!do_pts) A = 1;
  COATI_CHECK_SHAPE(B > 0 && T1 > 0 && A > 0 && T1 <= c.n_seq, "engine_encode: unsupported shape B=%d T1=%d A=%d", B, T1, A);
  hipStream_t s = (hipStream_t)stream;
  const int C = c.n_hidden_xformer, H = c.n_hidden_e3nn, E = c.n_embd_common;
  Arena ar{reinterpret_cast<char*>(workspace), 0, (size_t)workspace_bytes, false};
  const size_t need = carve(e, ar, B, T1, 1, A, B);
  COATI_CHECK_ARG((int64_t)need <= workspace_bytes, "engine_encode: workspace too small (%zu > %lld)", need, (long long)workspace_bytes);
  if (e->nce) e->nce_cap = ((size_t)workspace_bytes - (size_t)(reinterpret_cast<char*>(e->nce) - reinterpret_cast<char*>(workspace))) / sizeof(float);
  e->B = B; e->T1 = T1; e->T2 = 1; e->A = A;
  e->have_fwd = false;
  e->have_ws = true;
  HIPCHK(hipMemsetAsync(scal, 0, 16 * sizeof(float), s));
  HIPCHK(hipMemsetAsync(e->err_flag, 0, 4 * sizeof(int), s));
  if (do_pts) {
    e->atoms = reinterpret_cast<const long long*>(atoms);
    if (c.use_point_encoder) {
      COATI_TRY(gnn_fwd(e, e->atoms, coords, s));
      COATI_TRY(point_head_fwd(e, h_e3gnn, s));
    } else {
      HIPCHK(hipMemsetAsync(h_e3gnn, 0, (size_t)B * E * sizeof(float), s));   // clip_e2e.py:462-463
    }
  }
  if (do_tok) {
    e->p1.idx = reinterpret_cast<const long long*>(raw_tokens);
    COATI_TRY(xformer_fwd(e, e->p1, nullptr, s));
    COATI_TRY(launch_find_stop(e->p1.idx, c.stop_token, e->stop_pos, e->err_flag, B, T1, s));
    COATI_TRY(launch_gather_rows(e->p1.xf32, e->stop_pos, e->hstop, B, T1, C, s));
    COATI_TRY(smiles_head_fwd(e, h_smiles, s));
    HIPCHK(hipMemcpyAsync(scal + 6, e->err_flag, sizeof(int), hipMemcpyDeviceToDevice, s));
  }
  return COATI_OK;
}

int coati_engine_logits(coati_engine* e, float* logits, int64_t ldl, void* stream) {
  COATI_CHECK_ARG(e && e->have_fwd && logits, "engine_logits: no forward to read");
  const coati_config& c = e->cfg;
  hipStream_t s = (hipStream_t)stream;
  COATI_CHECK_ARG(!e->p2.packed, "engine_logits: the last forward ran on packed rows (logits of padded positions do not exist): run it with rows1 = rows2 = 0");
  return gemm(e, SITE_LMHEAD_FWD, e->p2.af, 0, c.n_hidden_xformer, e->S + e->lmhead, c.n_hidden_xformer, e->B * e->T2, c.n_tok,
              c.n_hidden_xformer, logits, ldl, nullptr, EPI_F32, nullptr, nullptr, 0, s);
}

int coati_engine_infonce(coati_engine* e, const float* S_loc, const float* C_loc, const float* S_all,
                         const float* C_all, const uint8_t* bad_all, int B, int Bg, int row0, float gscale,
                         float* dS_all, float* dC_all, float* scal, void* stream) {
  COATI_CHECK_ARG(e && S_loc && C_loc && S_all && C_all && bad_all && dS_all && dC_all && scal, "engine_infonce: null argument");
  COATI_CHECK_SHAPE(B > 0 && Bg >= B && row0 >= 0 && row0 + B <= Bg, "engine_infonce: bad row range");
  hipStream_t s = (hipStream_t)stream;
  const int E = e->cfg.n_embd_common;
  COATI_CHECK_ARG(e->have_ws, "engine_infonce: needs the workspace of a forward / encode call");
  // logits scratch [B, Bg] f32 x 2: the dedicated region sized at workspace time (world_size * B columns), or -- for the
  // stand-alone clip_loss API with another batch size -- the idle backward scratch of the decoder pass
  const size_t need = (size_t)2 * B * Bg;
  const size_t dh4_floats = (size_t)e->B * (e->T1 > e->T2 ? e->T1 : e->T2) * 4 * e->cfg.n_hidden_xformer * sizeof(bf16_t) / sizeof(float);
  COATI_CHECK_SHAPE(need <= e->nce_cap || need <= dh4_floats,
                    "engine_infonce: B=%d x Bg=%d does not fit the logits scratch (workspace was sized for Bg=%zu)", B, Bg,
                    e->nce_cap / (2 * (size_t)(e->B > 0 ? e->B : 1)));
  float* L1 = need <= e->nce_cap ? e->nce : reinterpret_cast<float*>(e->dh4);
  float* L2 = L1 + (size_t)B * Bg;
  COATI_TRY(launch_count_valid(bad_all, Bg, scal + 4, scal + 7, s));
  // L1 = S_loc C_all^T ; L2 = C_loc S_all^T     (clip_e2e.py:36-37, local rows only)
  {
    SgemmBatch sb;
    COATI_TRY(sgemm_batch_add(sb, S_loc, E, 1, C_all, 1, E, L1, Bg, B, Bg, E, nullptr, 1.f, 0));
    COATI_TRY(sgemm_batch_add(sb, C_loc, E, 1, S_all, 1, E, L2, Bg, B, Bg, E, nullptr, 1.f, 0));
    COATI_TRY(launch_sgemm_batch(sb, s));
  }
  COATI_TRY(launch_infonce_rows2(L1, L2, Bg, B, Bg, row0, bad_all, scal + 2, scal + 3, scal + 7, gscale, s));
  // column-side gradients: dC_all = dL1^T S_loc ; dS_all = dL2^T C_loc
  {
    SgemmBatch sb;
    COATI_TRY(sgemm_batch_add(sb, L1, 1, Bg, S_loc, E, 1, dC_all, E, Bg, E, B, nullptr, 1.f, 0));
    COATI_TRY(sgemm_batch_add(sb, L2, 1, Bg, C_loc, E, 1, dS_all, E, Bg, E, B, nullptr, 1.f, 0));
    COATI_TRY(launch_sgemm_batch(sb, s));
  }
  // row-side gradients added into the local row block: dS_loc += dL1 C_all ; dC_loc += dL2 S_all
  {
    SgemmBatch sb;
    COATI_TRY(sgemm_batch_add(sb, L1, Bg, 1, C_all, E, 1, dS_all + (size_t)row0 * E, E, B, E, Bg, nullptr, 1.f, 1));
    COATI_TRY(sgemm_batch_add(sb, L2, Bg, 1, S_all, E, 1, dC_all + (size_t)row0 * E, E, B, E, Bg, nullptr, 1.f, 1));
    COATI_TRY(launch_sgemm_batch(sb, s));
  }
  return COATI_OK;
}

int coati_engine_backward(coati_engine* e, const float* dh_smiles, const float* dh_e3gnn, int stage, void* stream) {
  COATI_CHECK_ARG(e && e->have_fwd && e->G, "engine_backward: no forward / gradient buffer");
  COATI_CHECK_ARG(stage >= 0 && stage <= 5, "engine_backward: bad stage");
  hipStream_t s = (hipStream_t)stream;
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer, H = c.n_hidden_e3nn, E = c.n_embd_common, B = e->B;
  if (stage == 0 || stage == 1) {
    const int M2 = e->p2.M;
    // ---- lm_head backward: dlogits (bf16) -> d(af), dW_lm ----
    {
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.A = e->p2.af; a.lda = C; a.B = e->S + e->lmhead; a.ldb = C; a.M = M2; a.N = c.n_tok; a.K = C;
      a.C = e->dlogits; a.ldc = e->Vpad; a.n_store = e->Vpad; a.lse = e->ce_lse; a.target = e->p2.packed ? e->p2.ypk : e->y_next; a.scal = e->scal;
      ProfScope ps(e, SITE_LMHEAD_DLOGITS, 2.0 * M2 * c.n_tok * C, s, (double)M2 * C * 2 + (double)c.n_tok * C * 2 + (double)M2 * e->Vpad * 2);
      COATI_TRY(gemm_rows(a, 0, EPI_CE_BWD, s));
    }
    COATI_TRY(wgrad(e, SITE_LMHEAD_WGRAD, e->dlogits, 0, e->Vpad, e->p2.af, C, M2, e->Vpad, C, e->G + e->lmhead, C, nullptr, c.n_tok, s));
    // ---- decoder pass (its first launch: the lm_head's input gradient d(af) = dlogits W, + ln_f's backward) ----
    HIPCHK(hipMemsetAsync(e->dcliptok, 0, (size_t)B * E * sizeof(float), s));
    COATI_TRY(xformer_bwd(e, e->p2, e->da, 0, e->dcliptok, s, -1, 0, e->dlogits));
    // dh = external (contrastive) gradient + the special-token path
    if (dh_e3gnn) HIPCHK(hipMemcpyAsync(e->dhe, dh_e3gnn, (size_t)B * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    else HIPCHK(hipMemsetAsync(e->dhe, 0, (size_t)B * E * sizeof(float), s));
    if (dh_smiles) HIPCHK(hipMemcpyAsync(e->dhs, dh_smiles, (size_t)B * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    else HIPCHK(hipMemsetAsync(e->dhs, 0, (size_t)B * E * sizeof(float), s));
    // ---- special-token head: cliptok = where(use_point, ptok, stok) ----
    if (c.token_mlp) {
      HIPCHK(hipMemsetAsync(e->dptok, 0, (size_t)B * E * sizeof(float), s));
      HIPCHK(hipMemsetAsync(e->dstok, 0, (size_t)B * E * sizeof(float), s));
      COATI_TRY(launch_select_rows_bwd(e->use_point, e->dcliptok, e->dptok, e->dstok, B, E, s));
      SgemmBatch sb;
      COATI_TRY(head_linear_bwd(e, sb, e->dptok, e->sa, e->tokw, e->tokb, e->dsa, B, E, E));
      COATI_TRY(head_linear_bwd(e, sb, e->dstok, e->sb, e->tokw, e->tokb, e->dsb, B, E, E));
      COATI_TRY(launch_sgemm_batch(sb, s));
      COATI_TRY(launch_silu_bwd(e->h_e3gnn, e->dsa, e->dhe, (long long)B * E, 1, s));
      COATI_TRY(launch_silu_bwd(e->h_smiles, e->dsb, e->dhs, (long long)B * E, 1, s));
    } else {
      // nn.Identity: the token IS the embedding, its gradient adds straight into d h_e3gnn / d h_smiles
      COATI_TRY(launch_select_rows_bwd(e->use_point, e->dcliptok, e->dhe, e->dhs, B, E, s));
    }
    // smiles_to_clip / point_to_clip: Linear then (norm_clips) LayerNorm backward; the two heads' Linear backwards share a launch
    const bool old_arch = c.norm_clips && c.old_architecture;
    if (old_arch) {
      // Linear -> LayerNorm: the LayerNorm's backward first (x = the Linear's output kept in hs_ln / hp_ln), its result in dhs_ln / dhp_ln
      COATI_TRY(launch_layernorm_bwd(e->dhs, 1, E, e->hs_ln, E, 0, e->hs_mean, e->hs_rstd, e->P + e->s2c_lnw, nullptr, e->dhs_ln, nullptr, e->G + e->s2c_lnw, e->G + e->s2c_lnb, e->ln_partial, B, E, s));
      if (c.use_point_encoder)
        COATI_TRY(launch_layernorm_bwd(e->dhe, 1, E, e->hp_ln, E, 0, e->hp_mean, e->hp_rstd, e->P + e->p2c_lnw, nullptr, e->dhp_ln, nullptr, e->G + e->p2c_lnw, e->G + e->p2c_lnb, e->ln_partial, B, E, s));
    }
    {
      SgemmBatch sb;
      if (old_arch) COATI_TRY(head_linear_bwd(e, sb, e->dhs_ln, e->hstop, e->s2c_w, e->s2c_b, e->dhstop, B, E, C));
      else if (c.norm_clips) COATI_TRY(head_linear_bwd(e, sb, e->dhs, e->hs_ln, e->s2c_w, e->s2c_b, e->dhs_ln, B, E, C));
      else COATI_TRY(head_linear_bwd(e, sb, e->dhs, e->hstop, e->s2c_w, e->s2c_b, e->dhstop, B, E, C));
      if (c.use_point_encoder) {   // (use_point_encoder = False: h_e3gnn is a constant, nothing upstream of it)
        if (old_arch) COATI_TRY(head_linear_bwd(e, sb, e->dhp_ln, e->hpoint, e->p2c_w, e->p2c_b, e->dhpoint, B, E, H));
        else if (c.norm_clips) COATI_TRY(head_linear_bwd(e, sb, e->dhe, e->hp_ln, e->p2c_w, e->p2c_b, e->dhp_ln, B, E, H));
        else COATI_TRY(head_linear_bwd(e, sb, e->dhe, e->hpoint, e->p2c_w, e->p2c_b, e->dhpoint, B, E, H));
      }
      COATI_TRY(launch_sgemm_batch(sb, s));
    }
    if (c.norm_clips && !old_arch) {
      COATI_TRY(launch_layernorm_bwd(e->dhs_ln, 1, C, e->hstop, C, 0, e->hs_mean, e->hs_rstd, e->P + e->s2c_lnw, nullptr, e->dhstop, nullptr, e->G + e->s2c_lnw, e->G + e->s2c_lnb, e->ln_partial, B, C, s));
      if (c.use_point_encoder)
        COATI_TRY(launch_layernorm_bwd(e->dhp_ln, 1, H, e->hpoint, H, 0, e->hp_mean, e->hp_rstd, e->P + e->p2c_lnw, nullptr, e->dhpoint, nullptr, e->G + e->p2c_lnw, e->G + e->p2c_lnb, e->ln_partial, B, H, s));
    }
  }
  // whole backward (stage 0) or the encoder stage of the staged (multi-GPU) backward: the point-encoder backward runs on
  // the side stream underneath the encoder pass; stage 3 then has nothing left to do
  // stages 4 / 5 = the encoder stage in two halves (upper / lower half of the layers), so that the caller can start the
  // all-reduce of the upper layers' finished gradients underneath the lower half
  const int Lx = c.n_layer_xformer, Lmid = Lx / 2;
  const bool ovl_bwd = (stage == 0 || stage == 2 || stage == 4) && e->overlap && (e->prof_mask == 0 || e->prof_keep_overlap) && c.use_point_encoder;
  if (stage == 0 || stage == 1) { e->gnn_bwd_done = false; e->gnn_side_pending = false; }
  if (ovl_bwd) {
    // the point-encoder backward only needs dhpoint (ready here) and writes its own gradient slice: side stream
    COATI_TRY(fork_side(e, s));
    COATI_TRY(gnn_bwd(e, e->dhpoint, e->side));
  }
  if (stage == 0 || stage == 2 || stage == 4) {
    // ---- encoder pass: gradient enters at the [STOP] rows of ln_f's output ----
    if (e->p1.tail) {   // the gradient of the B [STOP] rows goes in as it is
      COATI_TRY(xformer_bwd(e, e->p1, e->dhstop, 1, nullptr, s, Lx, stage == 4 ? Lmid : 0));
    } else {
      float* dxf = reinterpret_cast<float*>(e->dh4);  // [M1, C] f32 scratch (dh4 is idle here: 4C bf16 >= C f32)
      HIPCHK(hipMemsetAsync(dxf, 0, (size_t)e->p1.M * C * sizeof(float), s));
      COATI_TRY(launch_scatter_rows_add(e->dhstop, e->stop_pos, dxf, B, e->T1, C, s, e->p1.packed ? e->p1.off : nullptr));
      // xformer_bwd consumes dyf in its first kernel (ln_f backward) before dh4 is rewritten
      COATI_TRY(xformer_bwd(e, e->p1, dxf, 1, nullptr, s, Lx, stage == 4 ? Lmid : 0));
    }
  }
  if (stage == 5) COATI_TRY(xformer_bwd(e, e->p1, nullptr, 1, nullptr, s, Lmid, 0));
  if (ovl_bwd && stage == 4) {
    e->gnn_side_pending = true;   // joined at the end of stage 5
  } else if (ovl_bwd || (stage == 5 && e->gnn_side_pending)) {
    COATI_TRY(join_side(e, s));
    e->gnn_bwd_done = true;
    e->gnn_side_pending = false;
  } else if ((stage == 0 || stage == 3) && !e->gnn_bwd_done) {
    if (c.use_point_encoder) COATI_TRY(gnn_bwd(e, e->dhpoint, s));
    e->gnn_bwd_done = true;
  }
  return COATI_OK;
}

int coati_engine_optimizer_step(coati_engine* e, float lr, float beta1, float beta2, float eps, float weight_decay,
                                float max_norm, int step, float* scal, void* stream) {
  COATI_CHECK_ARG(e && e->P && e->G && e->Mo && e->Vo && scal, "optimizer_step: engine not bound for training");
  COATI_CHECK_ARG(e->have_fwd, "optimizer_step: needs the workspace of a forward call");
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps(e, SITE_OPTIM, 0, s);
    COATI_TRY(launch_grad_sqnorm(e->G, e->n_trainable, e->opt_partial, 1024, scal + 5, max_norm, scal + 8, s));
    COATI_TRY(launch_adamw(e->P, e->G, e->Mo, e->Vo, e->S, e->n_trainable, lr, beta1, beta2, eps, weight_decay, step, scal + 8, 1.f, s, e->err_flag));
  }
  return refresh_shadows_impl(e, stream, true);
}

// Data-parallel training: the step's error word (bit 0: a row without [STOP]; bit 1: packed-row counts that differ from what the
// device found) as reduced over ALL ranks replaces this rank's own one before coati_engine_optimizer_step, so that the AdamW kernel's
// `skip` drops the update on every rank or on none (the all-reduced gradient is the same everywhere: a rank that skipped alone
// would leave the replicas different).  word_dev: one int32 in device memory; also copied into scal[6] for losses().
int coati_engine_set_error_word(coati_engine* e, const int32_t* word_dev, void* stream) {
  COATI_CHECK_ARG(e && word_dev && e->have_ws && e->err_flag && e->scal, "set_error_word: no forward workspace");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(e->err_flag, word_dev, sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess ||
      hipMemcpyAsync(e->scal + 6, word_dev, sizeof(int), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    coati_set_error("set_error_word: copy failed");
    return COATI_EHIP;
  }
  return COATI_OK;
}

int coati_engine_prof_select(coati_engine* e, int site) {
  COATI_CHECK_ARG(e && site >= -1 && site < SITE_COUNT, "prof_select: bad site");
  if (site >= 0 && e->ev.empty()) {
    e->ev.resize(8192);
    for (auto& ev : e->ev) {
      if (hipEventCreate(&ev) != hipSuccess) {
        coati_set_error("prof_select: hipEventCreate failed");
        return COATI_EHIP;
      }
    }
  }
  e->prof_mask = site >= 0 ? (1ull << site) : 0ull;
  e->prof_keep_overlap = false;
  e->prof_paused = false;
  e->ev_used = 0;
  e->prof_flops = 0.0;
  e->prof_bytes = 0.0;
  return COATI_OK;
}

// several launch sites at once (a kernel that serves more than one site, e.g. the ring GEMM with the LayerNorm backward in its
// write-out = fc1_dgrad + qkv_dgrad): prof_select first (it creates the events), then OR further sites in
int coati_engine_prof_add_site(coati_engine* e, int site) {
  COATI_CHECK_ARG(e && site >= 0 && site < SITE_COUNT && e->prof_mask != 0, "prof_add_site: bad site / nothing selected");
  e->prof_mask |= 1ull << site;
  return COATI_OK;
}

// keep != 0: the selected site is timed while the step runs exactly as the product runs it (point encoder on the side stream,
// concurrent with the transformer passes); the default (0) serialises the point encoder onto the launch stream so that a
// site's events bracket that site's kernels alone.  Reset by every prof_select.
// paused != 0: the selected sites record no events until resumed (selection, counters and the overlap setting stay).  A pair of
// HIP events costs ~ 3.7 us of queue time on this chip (a barrier packet each: 0.48 ms per step for the 64 launches of the ring GEMM
// with the LayerNorm backward, tools/dp_overhead.py), so the bench samples a quarter of its timed steps.
int coati_engine_prof_pause(coati_engine* e, int paused) {
  COATI_CHECK_ARG(e, "prof_pause: null engine");
  e->prof_paused = paused != 0;
  return COATI_OK;
}

int coati_engine_prof_keep_overlap(coati_engine* e, int keep) {
  COATI_CHECK_ARG(e, "prof_keep_overlap: null engine");
  e->prof_keep_overlap = keep != 0;
  return COATI_OK;
}

int coati_engine_prof_collect(coati_engine* e, double* total_ms, int64_t* launches, double* flops_per_launch) {
  COATI_CHECK_ARG(e && total_ms && launches, "prof_collect: null argument");
  double tot = 0.0;
  for (int i = 0; i + 1 < e->ev_used; i += 2) {
    hipEventSynchronize(e->ev[i + 1]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]) != hipSuccess) {
      coati_set_error("prof_collect: hipEventElapsedTime failed");
      return COATI_EHIP;
    }
    tot += ms;
  }
  *total_ms = tot;
  *launches = e->ev_used / 2;
  if ((e->prof_bytes_e > 0.0 || e->prof_flops_e > 0.0) && e->have_ws && e->g_ne != nullptr) {
    int ne = 0;   // edges of the last step's neighbour list (every launch above has completed: the events were waited for)
    if (hipMemcpy(&ne, e->g_ne, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && ne > 0) {
      e->prof_bytes += e->prof_bytes_e * ne;
      e->prof_flops += e->prof_flops_e * ne;
    }
  }
  if (flops_per_launch) *flops_per_launch = e->ev_used ? e->prof_flops / (e->ev_used / 2) : 0.0;
  e->prof_last_bytes = e->ev_used ? e->prof_bytes / (e->ev_used / 2) : 0.0;
  e->ev_used = 0;
  e->prof_flops = 0.0;
  e->prof_bytes = 0.0;
  e->prof_bytes_e = 0.0;
  e->prof_flops_e = 0.0;
  return COATI_OK;
}

int coati_engine_prof_last_bytes(coati_engine* e, double* bytes_per_launch) {
  COATI_CHECK_ARG(e && bytes_per_launch, "prof_last_bytes: null argument");
  *bytes_per_launch = e->prof_last_bytes;
  return COATI_OK;
}

int coati_engine_site_count(void) { return SITE_COUNT; }
const char* coati_engine_site_name(int site) { return (site >= 0 && site < SITE_COUNT) ? kSiteNames[site] : ""; }

}  // extern "C"

namespace { void decode_drop_graphs(coati_engine* e); }

// ---- inference: KV-cached decode (reference smiles_xformer.py:272-351 + xformer_blocks) ----------------
namespace {
size_t decode_carve(coati_engine* e, Arena& ar, int B, int Tmax) {
  const coati_config& c = e->cfg;
  const size_t C = c.n_hidden_xformer, L = c.n_layer_xformer;
  auto& d = e->dec;
  d.cache = ar.take<bf16_t>(L * B * (size_t)C * Tmax * 2);   // [L][B][nh][Tmax][k | v]
  d.x = ar.take<float>(B * C); d.xmid = ar.take<float>(B * C); d.xn = ar.take<float>(B * C);
  d.mean = ar.take<float>(B); d.rstd = ar.take<float>(B);
  d.a = ar.take<bf16_t>(B * C); d.qkv = ar.take<bf16_t>(B * 3 * C); d.y = ar.take<bf16_t>(B * C);
  d.hpre = ar.take<bf16_t>(B * 4 * C); d.g = ar.take<bf16_t>(B * 4 * C); d.af = ar.take<bf16_t>(B * C);
  d.pos_dev = ar.take<int>(4);
  d.tok_dev = ar.take<long long>(B);
  d.inj_dev = ar.take<float>(B * C);
  d.ldl = ((int64_t)c.n_tok + 7) / 8 * 8;
  d.logits_dev = ar.take<float>((size_t)B * d.ldl);
  return (ar.off + 255) & ~(size_t)255;
}
}  // namespace

extern "C" {

int64_t coati_engine_decode_workspace_bytes(coati_engine* e, int B, int Tmax) {
  if (!e || B <= 0 || Tmax <= 0) return 0;
  Arena ar{nullptr, 0, 0, true};
  coati_engine::Decode keep = e->dec;
  const size_t n = decode_carve(e, ar, B, Tmax);
  e->dec = keep;
  return (int64_t)n;
}

int coati_engine_decode_begin(coati_engine* e, void* workspace, int64_t ws_bytes, int B, int Tmax) {
  COATI_CHECK_ARG(e && workspace && e->P && e->S, "decode_begin: engine not bound / null workspace");
  COATI_CHECK_SHAPE(B > 0 && Tmax > 0 && Tmax <= e->cfg.n_seq && Tmax <= 256, "decode_begin: bad shape B=%d Tmax=%d (n_seq=%d)", B, Tmax, e->cfg.n_seq);
  COATI_CHECK_SHAPE(ws_bytes >= coati_engine_decode_workspace_bytes(e, B, Tmax), "decode_begin: workspace too small");
  decode_drop_graphs(e);   // a captured graph bakes the previous session's buffer addresses in
  Arena ar{reinterpret_cast<char*>(workspace), 0, (size_t)ws_bytes, false};
  decode_carve(e, ar, B, Tmax);
  e->dec.active = true;
  e->dec.B = B; e->dec.Tmax = Tmax; e->dec.pos = 0;
  return COATI_OK;
}

int coati_engine_decode_pos(coati_engine* e) { return (e && e->dec.active) ? e->dec.pos : -1; }

}  // extern "C"

namespace {
// Enqueue one decode position.  graph_mode: the position comes from device memory (d.pos_dev) so that the very same
// launch sequence can be replayed from a captured graph, and the sequence ends by incrementing it.
int decode_enqueue(coati_engine* e, const long long* tokens, const float* injection, float* logits, int64_t ldl,
                   bool graph_mode, hipStream_t s) {
  auto& d = e->dec;
  const coati_config& c = e->cfg;
  const int C = c.n_hidden_xformer, L = c.n_layer_xformer, B = d.B, hs = C / c.n_head;
  if (c.norm_embed) {
    COATI_TRY(launch_embed_fwd(tokens, e->P + e->tok_emb, nullptr, c.unk_token, d.xmid, B, 1, C, c.n_tok, s));   // (xmid: free until the first block writes it)
    COATI_TRY(launch_layernorm_fwd(d.xmid, C, e->P + e->emb_lnw, e->P + e->emb_lnb, nullptr, 0, d.x, C, d.mean, d.rstd, B, C, s));
    if (injection != nullptr) COATI_TRY(launch_embed_fwd(tokens, nullptr, injection, c.unk_token, d.x, B, 1, C, c.n_tok, s, nullptr, 0, 1));
  } else {
    COATI_TRY(launch_embed_fwd(tokens, e->P + e->tok_emb, injection, c.unk_token, d.x, B, 1, C, c.n_tok, s));
  }
  float* x = d.x;
  float* xm = d.xmid;
  for (int l = 0; l < L; ++l) {
    const XLayerP& w = e->xl[l];
    COATI_TRY(launch_layernorm_fwd(x, C, e->P + w.ln1w, e->P + w.ln1b, d.a, C, nullptr, 0, d.mean, d.rstd, B, C, s));
    {
      GemmArgs a;
      memset(&a, 0, sizeof(a));
      a.A = d.a; a.lda = C; a.B = e->S + w.attnw; a.ldb = C; a.M = B; a.N = 3 * C; a.K = C; a.C = d.qkv; a.ldc = 3 * C;
      a.bias = e->P + w.attnb; a.rope_hs = hs; a.rope_T = 1; a.rope_C = C;
      if (graph_mode) {   // every row sits at token position *pos_dev
        a.rope_cos = e->cos_t; a.rope_sin = e->sin_t; a.rope_pos = d.pos_dev;
      } else {            // every row sits at token position pos: tables offset to that row, period 1
        a.rope_cos = e->cos_t + (size_t)d.pos * hs; a.rope_sin = e->sin_t + (size_t)d.pos * hs;
      }
      COATI_TRY(launch_gemm_nt(a, 0, EPI_QKV_ROPE, s));
    }
    bf16_t* cache_l = d.cache + (size_t)l * B * C * d.Tmax * 2;
    COATI_TRY(launch_attn_decode(d.qkv, cache_l, d.y, B, c.n_head, hs, d.Tmax, d.pos, graph_mode ? d.pos_dev : nullptr, s));
    COATI_TRY(gemm(e, SITE_NONE, d.y, 0, C, e->S + w.projw, C, B, C, C, xm, C, e->P + w.projb, EPI_RES_F32, x, nullptr, C, s));
    COATI_TRY(launch_layernorm_fwd(xm, C, e->P + w.ln2w, e->P + w.ln2b, d.a, C, nullptr, 0, d.mean, d.rstd, B, C, s));
    COATI_TRY(gemm(e, SITE_NONE, d.a, 0, C, e->S + w.fc1w, C, B, 4 * C, C, d.g, 4 * C, e->P + w.fc1b, EPI_GELU, nullptr, d.hpre, 4 * C, s));
    COATI_TRY(gemm(e, SITE_NONE, d.g, 0, 4 * C, e->S + w.fc2w, 4 * C, B, C, 4 * C, x, C, e->P + w.fc2b, EPI_RES_F32, xm, nullptr, C, s));
  }
  if (logits) {
    COATI_TRY(launch_layernorm_fwd(x, C, e->P + e->lnfw, e->P + e->lnfb, d.af, C, nullptr, 0, d.mean, d.rstd, B, C, s));
    COATI_TRY(gemm(e, SITE_NONE, d.af, 0, C, e->S + e->lmhead, C, B, c.n_tok, C, logits, ldl, nullptr, EPI_F32, nullptr, nullptr, 0, s));
  }
  if (graph_mode) COATI_TRY(launch_add_int(d.pos_dev, 1, 0, s));
  return COATI_OK;
}

void decode_drop_graphs(coati_engine* e) {
  for (auto& g : e->dec.graph) {
    if (g) hipGraphExecDestroy(g);
    g = nullptr;
  }
}
}  // namespace

extern "C" {

// One position for every sequence: tokens[B] (ids; rows equal to the [UNK] id take their embedding from injection[B, C]
// when it is given, smiles_xformer.py:444-448).  logits (optional) [B, n_tok] f32, row stride ldl.
int coati_engine_decode_step(coati_engine* e, const int64_t* tokens, const float* injection, float* logits, int64_t ldl,
                             void* stream) {
  COATI_CHECK_ARG(e && e->dec.active && tokens, "decode_step: no decode session / null tokens");
  auto& d = e->dec;
  COATI_CHECK_SHAPE(d.pos < d.Tmax, "decode_step: the cache is full (pos=%d, Tmax=%d)", d.pos, d.Tmax);
  COATI_CHECK_ARG(!logits || ldl >= e->cfg.n_tok, "decode_step: ldl too small");
  COATI_TRY(decode_enqueue(e, reinterpret_cast<const long long*>(tokens), injection, logits, ldl, false, (hipStream_t)stream));
  d.pos += 1;
  return COATI_OK;
}

// Capture the decode step (embedding .. logits + position increment, ~115 launches) into two HIP graphs (without / with
// the [UNK]-slot injection).  The launch-bound small-batch step then costs one hipGraphLaunch.  Call after decode_begin;
// a dry, un-captured step runs first so that every one-time kernel attribute is set outside the capture (it writes the
// cache slot of the CURRENT position, which the next real step overwrites).
int coati_engine_decode_graph_build(coati_engine* e, void* stream) {
  COATI_CHECK_ARG(e && e->dec.active, "decode_graph_build: no decode session");
  auto& d = e->dec;
  hipStream_t s = (hipStream_t)stream;
  COATI_CHECK_ARG(s != nullptr, "decode_graph_build: graph capture needs an explicit (non-default) stream");
  decode_drop_graphs(e);
  HIPCHK(hipMemsetAsync(d.tok_dev, 0, sizeof(long long) * d.B, s));
  HIPCHK(hipMemsetAsync(d.inj_dev, 0, sizeof(float) * d.B * e->cfg.n_hidden_xformer, s));
  COATI_TRY(decode_enqueue(e, d.tok_dev, d.inj_dev, d.logits_dev, d.ldl, false, s));
  HIPCHK(hipStreamSynchronize(s));
  for (int v = 0; v < 2; ++v) {
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    const int rc = decode_enqueue(e, d.tok_dev, v ? d.inj_dev : nullptr, d.logits_dev, d.ldl, true, s);
    hipError_t ee = hipStreamEndCapture(s, &g);
    if (rc != COATI_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (ee != hipSuccess || !g) {
      coati_set_error("decode_graph_build: capture failed: %s", hipGetErrorString(ee));
      return COATI_EHIP;
    }
    ee = hipGraphInstantiate(&d.graph[v], g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (ee != hipSuccess) {
      coati_set_error("decode_graph_build: hipGraphInstantiate failed: %s", hipGetErrorString(ee));
      return COATI_EHIP;
    }
  }
  return COATI_OK;
}

// Replay: tokens[B] / injection[B, C] are copied to the graph's fixed input buffers, the graph runs one position and
// leaves the logits in the session's own buffer (*logits_out, row stride *ldl_out; valid until the next step).
int coati_engine_decode_graph_step(coati_engine* e, const int64_t* tokens, const float* injection, float** logits_out,
                                   int64_t* ldl_out, void* stream) {
  COATI_CHECK_ARG(e && e->dec.active && tokens && logits_out && ldl_out, "decode_graph_step: no decode session / null argument");
  auto& d = e->dec;
  COATI_CHECK_ARG(d.graph[0] && d.graph[1], "decode_graph_step: call coati_engine_decode_graph_build first");
  COATI_CHECK_SHAPE(d.pos < d.Tmax, "decode_graph_step: the cache is full (pos=%d, Tmax=%d)", d.pos, d.Tmax);
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(d.tok_dev, tokens, sizeof(long long) * d.B, hipMemcpyDeviceToDevice, s));
  if (injection) HIPCHK(hipMemcpyAsync(d.inj_dev, injection, sizeof(float) * d.B * e->cfg.n_hidden_xformer, hipMemcpyDeviceToDevice, s));
  COATI_TRY(launch_add_int(d.pos_dev, d.pos, 1, s));   // keeps the device position in step with the host's (eager steps may have run)
  HIPCHK(hipGraphLaunch(d.graph[injection ? 1 : 0], s));
  d.pos += 1;
  *logits_out = d.logits_dev;
  *ldl_out = d.ldl;
  return COATI_OK;
}

}  // extern "C"
