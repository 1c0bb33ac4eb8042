// persist.hip.h — the latency-bound three quarters of a single utterance as PERSISTENT step programs (round 3).
//
// What it replaces: at B = 1 the text encoder, the stochastic duration predictor and the flow are ~120 dependent launches on
// [192..768 x 50..160] tensors that carry 21 % of the forward's FLOPs and take 0.9 of its 1.33 ms: every launch pays the dispatch, a
// cold L2 and its own chain of dependent cold misses, 5.5-13 us each.
//
// How: a stage is a STEP PROGRAM run by ONE kernel of P workgroups (one per CU) that never leave the machine.  Between steps there
// is NO barrier and NO flag: every exchanged tensor is an array of 8-byte "LL cells" {float value, u32 epoch} written with one
// agent-scope 8-byte store and polled by the consumers with agent-scope (L1-bypassing, sc1) 8-byte loads until the epoch matches
// this forward -- the data is its own arrival signal (tools/llprobe.hip: 0.32 us one-way inside an XCD, 0.73 us across XCDs,
// coherent chip-wide with sc1 stores).  A worker that is idle in a step simply moves on; a consumer waits only for the cells it
// reads.  Epochs make stale data harmless: every forward uses epoch = (last completed forward) + 1, cells are never reset, every step
// of a forward writes its OWN buffers (no reuse inside a forward, so there is no write-after-read hazard either), and a cell whose
// epoch does not match is simply not there yet.  Every poll loop is bounded (PS_SPIN_LIMIT): a lost worker turns into an error
// word, never into a hung GPU.
//
// What bounds a step (measured, tools/ps_trace.py): (1) how many bytes ONE workgroup must pull through L1-bypassing loads (~30 GB/s
// per CU), (2) how often an element is recomputed, (3) the INSTRUCTIONS every wave issues for bookkeeping (a wave issues at most one
// instruction per 4 cycles: a generic interpreter that decoded its step descriptors on the device ran 15.6 k cycles per matrix step
// against 6.1 k for hand-specialised code).  So:
//   * the HOST resolves everything: per (step, worker) one 128-byte RECORD holds the final pointers of that worker's work item
//     (operand window, weight fragments of its row blocks, epilogue operands, output tile) and a few packed integers; the kernel
//     keeps the record in 4 VGPRs of lanes 0..7 and reads fields with v_readlane -- no divisions, no clamping, no descriptor
//     decode on the device; records and read-only operands are requested two / one steps ahead in straight-line code;
//   * column steps, one worker per COLUMN t of the column-major cells [T][C], thread = channel: LayerNorm (+ K-slice partial
//     sums, bias, residual, speaker vector), the DDSConv elementwise chain (finish the previous layer at t - d, t, t + d,
//     depthwise conv, LN, GELU), embedding lookup, attention merge, coupling tail.  A LayerNorm is a DPP wave reduction + one LDS
//     exchange; every element is computed once;
//   * matrix steps, one worker per (16-column tile, group of 16-row blocks, K-slice): gather the [C_in x (16 + taps - 1)] operand
//     window (<= 40 KB), 16x16x4 fp32 MFMAs with the 8 waves splitting the contraction at tap-unit granularity, epilogue (bias /
//     per-item bias / ReLU / mask / residual, WaveNet gate, spline inverse), cells out.  Contractions over 768 channels are split
//     into K-slices whose partial sums the consuming LayerNorm / coupling column step adds up;
//   * attention = 16 x 16 (query tile, key tile) block steps (each gathers three 16 x d_k tiles, partial softmax with the banded
//     relative-position terms of attentions.py:165-260) + a merge column step.
//   * a step is whatever records the resolver wrote for it: sequences longer than the worker count (T up to PS_MAX_T = 512 with
//     P = 256 workers) run a column step as two rounds of P columns and an attention step as rounds of P blocks -- the kernel is
//     the same (round 4; DESIGN.md "Programs beyond 256 columns").
// (v1 of the duration predictor ran a whole DDSConv layer per step like conv16_kernel's PRO == 1: every one of the 16 workgroups of
// a column tile pulled the full 256-channel x 34-column window of two tensors and redid its LayerNorms and 8 k erf evaluations:
// 12 us per step, 208 us for the predictor against 265 us of launches.)
// Reference ops: attentions.py:48-65,133-260,292-317 (Encoder / MultiHeadAttention / FFN), models.py:56-63,93-101 (duration
// predictor), modules.py:96-108,148-176,363-390 (DDSConv, WN, ConvFlow), models.py:374-393 (coupling layer).
#pragma once
#include "conv_small.hip.h"
#include "kernels_misc.hip.h"

typedef unsigned long long ll_t;  // {float value (bits 0..31), u32 epoch (bits 32..63)}

#define PS_THREADS 512
#define PS_WAVES 8
#define PS_MAX_STEPS 320
#define PS_MAX_T 512         // longest sequence of a program (columns beyond the worker count run as further rounds of a step)
#define PS_MAXC 256      // channels of a column step / contraction channels of one K-slice
#define PS_MAXU 8        // tap units (16 channels x 1 tap) per wave
#define PS_TP 21         // LDS pitch of the MFMA operand window [C_in][16 + taps - 1] (odd: the transposing writes are conflict-free)
#define PS_MAXROW 20     // 16 + taps - 1, taps <= 5
#define PS_DKP 128       // attention: head-dimension slots per row of threads (d_k <= 128)
#define PS_MKT 12        // key tiles the attention merge polls per round (3 cells each)
#define PS_SPIN_LIMIT (1 << 18)
#define PS_TUNE_TOUCH 1   // PCall.tune bits
#define PS_TUNE_DEFAULT 0
#define PS_ERR_TIMEOUT 8  // bit in the session error word

struct PersistCtl {
  unsigned epoch;     // epoch of the last completed forward
  unsigned done;      // workers that finished the current forward (the last one publishes the epoch and resets this)
  unsigned abort;     // a worker timed out: everybody stops polling
  unsigned timeouts;  // diagnostics
};

enum { PK_IDLE = 0, PK_MM = 1, PK_DDS = 2, PK_LN = 3, PK_EMB = 4, PK_ATT = 5, PK_MERGE = 6, PK_COUPLE = 7, PK_DUR = 8, PK_EXPAND = 9 };
// flags (record dword 0, bits 8..)
enum {
  PF_RELU = 1 << 8, PF_INMASK = 1 << 9, PF_OUTMASK = 1 << 10, PF_GATE = 1 << 11, PF_SPLINE = 1 << 12, PF_ZINIT = 1 << 13, PF_LAST = 1 << 14,
  PF_PLAIN_IN = 1 << 15,   // PK_MM: operand = plain floats [channels][pT]; PK_COUPLE: previous z = plain floats
  PF_DW = 1 << 16, PF_FIN_LN = 1 << 17, PF_FIN_PRE = 1 << 18,  // PK_DDS
  PF_LN = 1 << 19,         // PK_LN: normalise (else: plain sum)
};

// One work item of one worker in one step, resolved on the host (persist_plan.hip.h): 32 dwords.
//   dword 0      kind | flags
//   dwords 1..3  a1..a3 (kind-specific integers)
//   dwords 4..23 p0..p9 (pointers; p6 = weight fragments, p7 = bias, p8 = per-item bias / vector, p9 = packed per-thread parameters:
//                these four are valid in EVERY record -- zeros when unused -- so that the prefetch is straight-line code)
//   dwords 24..31 b0..b7 (kind-specific integers)
// PK_MM    a1 = Cs | K << 16 ; a2 = cell pitch of the operand * channel direction (signed; plain operands: +-1); a3 = t0 (first window column)
//          p0 operand (cells: (column 0, LOWEST channel of the slice); plain: the row of the slice's first channel),
//          p1 yout tile (cell (n0, first row)), p2 yplain (first row, column n0), p3 res tile, p4 z, p5 zout
//          p9 = weight fragments of this worker's NEXT matrix item (first row block; 0: none)
//          b0 = n_u | nblk << 7 | next item's (n_u << 11 | nblk << 18 | KB between its row blocks << 22);
//          b1 = floats between consecutive row blocks of the weights; b2 = ypitch; b3 = rows left (Cout - first row);
//          b4 = pT; b5 = rpitch; b6 = n0 | z_row << 16 | ea_row << 20; b7 = gate_H
// PK_DDS   a1 = C | dil << 16; a2 = t; p0 xin, p1 y2, p2 z row, p3 xout, p4 bout
// PK_LN    a1 = C; a2 = t; a3 = np; p0 part, p1 res, p2 base, p3 out, p4 oplain; b0/b1 = part stride (cells, 64 bit); b4 = pT
// PK_EMB   a1 = C; a2 = t; p0 emb, p3 out, p4 oplain; b0 = scale (float bits); b1 = n_vocab; b4 = pT
// PK_ATT   a1 = dk | nh << 8 | W << 16; a2 = i0 | j0 << 16; a3 = head | key tile << 8; p0 qkv, p1 ap
// PK_MERGE a1 = C; a2 = t; a3 = dk | nh << 8; p0 ap, p3 out; b0 = cells per key tile
// PK_COUPLE a1 = C (= 2H); a2 = t; a3 = np | H << 16; p0 part, p1 u cells, p2 u plain, p3 out, p4 oplain; b0/b1 = part stride; b4 = pT
// PK_DUR   (one worker) a1 = T_x; a2 = frame capacity (0: none); flag PF_PLAIN_IN: logw is not waited for (no duration predictor in this
//          program); p0 logw cells [T_x], p1 dur (plain int), p2 cum (plain int), p3 cum cells, p4 y_len int32, p5 y_len int64, p6 y_len cell
// PK_EXPAND a1 = I; a2 = frame f; a3 = T_x; flag PF_PLAIN_IN: stats / cum are plain arrays of an earlier launch; p0 stats cells [T_x][2I] or
//          plain [2I][T_x], p1 cum cells or plain int [T_x], p3 out cells [T_y][I], p4 oplain [I][pT]; b4 = pT
struct alignas(16) PRec {
  int kf, a1, a2, a3;
  unsigned long long p[10];
  int b[8];
};
static_assert(sizeof(PRec) == 128, "PRec is 32 dwords");

struct PProgram {
  int n_steps, T, Tp, P;
  int nb; float bound, inv_sqrt_d;      // spline
  const int* len;
  const float* ea_m; const float* ea_logs;
  float* logw;                          // duration predictor result [T] (plain floats)
  int* err;
  const PRec* recs;                     // [n_steps][P]
  // programs that cover both halves of the forward (text side laid out for T_x, frame side for T_y): from step seg_step on the kernel
  // works with (T2, Tp2) and the utterance's frame count, which the PK_DUR step of the SAME launch publishes in the cell len2 (-1: one half only)
  int seg_step, T2, Tp2;
  const ll_t* len2;
  ll_t* logw_cells;                     // duration predictor result as cells for PK_DUR (null: plain only)
};

struct PCall {                          // per-call values (by value: a captured graph re-reads `dv`, not these, when dv != null)
  PersistCtl* ctl;
  const long long* ids;                 // text encoder: token ids [T] (the feed's int64 tensor)
  const float* noise;                   // duration predictor: [2][T] injected noise or null (Philox)
  float nsw;
  unsigned long long seed;
  int solo;
  const SynthDev* dv;
  const unsigned long long* item_seeds;
  long long* trace;                     // tools only (VITS_PS_TRACE): [P][PS_MAX_STEPS][8] cycle stamps, null in production
  const int* forced;                    // PK_DUR: pinned durations [T_x] (null: from logw)
  float length_scale, noise_scale;      // PK_DUR / PK_EXPAND (overridden by dv)
  const float* noise_prior;             // PK_EXPAND: injected prior noise [I][noise_stride] or null (Philox stream 2)
  long long noise_stride;
  int tune;                             // experiment switches (VITS_PS_TUNE; default PS_TUNE_DEFAULT): bit 0 = pull the next matrix step's weights into L2 ahead of time
  int* dbg;                             // device words of the model: [0] poll rounds before a worker gives up (0 = PS_SPIN_LIMIT; tests shrink it to
                                        // force the fallback -- read at run time, so captured graphs follow the hook), [1] completed persistent launches
};

#define PS_G __attribute__((address_space(1)))
__device__ __forceinline__ ll_t ll_pack(float v, unsigned e) { return ((ll_t)e << 32) | (ll_t)__float_as_uint(v); }
__device__ __forceinline__ void ll_store(PS_G ll_t* p, float v, unsigned e) {
  __hip_atomic_store(p, ll_pack(v, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 ... sc1
}
__device__ __forceinline__ ll_t ll_load(const PS_G ll_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 ... sc1
}
__device__ __forceinline__ ll_t ll_load_off(const PS_G ll_t* base, unsigned byte_off) {
  return ll_load((const PS_G ll_t*)((const PS_G char*)base + byte_off));
}
__device__ __forceinline__ void ll_store_off(PS_G ll_t* base, unsigned byte_off, float v, unsigned e) {  // uniform base + 32-bit lane offset
  ll_store((PS_G ll_t*)((PS_G char*)base + byte_off), v, e);
}
__device__ __forceinline__ float ll_val(ll_t q) { return __uint_as_float((unsigned)q); }
__device__ __forceinline__ unsigned ll_bad(ll_t q, unsigned epoch) { return (unsigned)(q >> 32) ^ epoch; }

// a record lives in 4 VGPRs (dwordx4 of lanes 0..7); fields are read with v_readlane (wave-uniform results in SGPRs)
typedef int ps_i4 __attribute__((ext_vector_type(4)));
#define PR_I(rv, i) __builtin_amdgcn_readlane((rv)[(i) & 3], (i) >> 2)
#define PR_P(T, rv, k) ((PS_G T*)(((unsigned long long)(unsigned)PR_I(rv, 5 + 2 * (k)) << 32) | (unsigned long long)(unsigned)PR_I(rv, 4 + 2 * (k))))
#define PR_B(rv, k) PR_I(rv, 24 + (k))

#define PS_GELU(v) c16_gelu(v)

struct PsCtx {
  unsigned epoch;
  int aborted;        // this wave gave up (or saw ctl->abort): polls return at once
  int spins;
  int limit;
  PersistCtl* ctl;
};
// one more round of a poll loop: true = keep polling.  `pending` is wave-uniform (a ballot).
__device__ __forceinline__ bool ps_again(PsCtx& cx, bool pending) {
  // The poll state is wave-uniform by construction, but hipcc's uniformity analysis gives up on it (it is carried around the step
  // loop through every kind's control flow) and kept `aborted` / `spins` as per-lane values: the exit test of every poll loop was a
  // vector compare under an exec mask.  readfirstlane at the point of use puts the test back on the scalar unit.
  if (!pending || __builtin_amdgcn_readfirstlane(cx.aborted)) { cx.spins = 0; return false; }
  const int spins = __builtin_amdgcn_readfirstlane(cx.spins) + 1;
  cx.spins = spins;
  // a poll that has failed a few rounds is waiting for something that is far away (another stage of the program, a slow worker): back
  // off instead of hammering the fabric next to the workers that are computing (MI355X_MICROARCH.md "polling-cost")
  if (spins > 8) __builtin_amdgcn_s_sleep(8);
  if ((spins & 1023) == 0 &&
      __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cx.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { cx.aborted = 1; return false; }
  if (spins >= cx.limit) {
    cx.aborted = 1;
    if ((threadIdx.x & 63) == 0) {
      __hip_atomic_store(&cx.ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(&cx.ctl->timeouts, 1u);
    }
    return false;
  }
  return true;
}
#define PS_PENDING(bad) (__builtin_amdgcn_ballot_w64((bad) != 0) != 0)

// sum / max over lanes (DPP row operations + two row broadcasts: no LDS, ~10 instructions).  Lanes disabled by ROW_MASK contribute 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ps_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ float ps_dpp_max(float v) {  // row-local controls only: every lane has a source
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float ps_row_sum(float v) {  // sum over each 16-lane row, in every lane of the row
  v = ps_dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = ps_dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = ps_dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = ps_dpp_add<0x140, 0xF>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ float ps_row_max(float v) {
  v = ps_dpp_max<0xB1>(v);
  v = ps_dpp_max<0x4E>(v);
  v = ps_dpp_max<0x141>(v);
  v = ps_dpp_max<0x140>(v);
  return v;
}
template <int N>
__device__ __forceinline__ float ps_row_shr(float v) {  // lane k <- lane k - N of its 16-lane row (0 where the row has no such lane)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xF, 0xF, true));
}
__device__ __forceinline__ float ps_wave_sum(float v) {  // sum over the 64 lanes, in every lane
  v = ps_row_sum(v);
  v = ps_dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v = ps_dpp_add<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sums of two values over the 256 threads of each HALF of the workgroup (waves 0-3 / 4-7): one barrier; `red` is a private
// 16-float scratch of this call site (no second barrier: the next call site uses another one)
__device__ __forceinline__ void ps_half_sum2(float& a, float& b, float* red, int wave, int lane) {
  a = ps_wave_sum(a);
  b = ps_wave_sum(b);
  if (lane == 0) { red[wave] = a; red[8 + wave] = b; }
  __syncthreads();
  const int w0 = wave & 4;
  a = (red[w0] + red[w0 + 1]) + (red[w0 + 2] + red[w0 + 3]);
  b = (red[8 + w0] + red[8 + w0 + 1]) + (red[8 + w0 + 2] + red[8 + w0 + 3]);
}

// ---- what a worker requests one step ahead (registers; straight-line: a load behind a branch makes hipcc wait for it at the join,
//      i.e. puts a cold round trip in the middle of a step)
struct PsPre {
  f32x4 a[PS_MAXU];   // weight fragments of the first 16-row block: tap units u = wave + PS_WAVES * i
  f32x4 pk0, pk1;     // packed per-thread parameters: 8 floats of row (tid & 255) -- attention: row tid of the table pack
  float eb0, eb1, ec0, ec1;  // epilogue operands of thread tid < 256 (bias, per-item bias / vector; second pair: the sigmoid row of a gate)
};
// LDS byte offset of tap unit u = wave + 8 i inside the operand window [C_in][PS_TP] for K = 1 / 3 / 5 taps: unit u = (16-channel
// chunk u / K, tap u % K) -> (chunk * 16 * PS_TP + tap) * 4.  One s_load_dwordx8 per step instead of a division-free but
// 8-instruction (chunk, tap) update per unit per wave.
struct PsUnitTab { int off[3][PS_WAVES][PS_MAXU]; };
static constexpr PsUnitTab ps_make_unit_tab() {
  PsUnitTab t{};
  for (int ki = 0; ki < 3; ++ki)
    for (int w = 0; w < PS_WAVES; ++w)
      for (int i = 0; i < PS_MAXU; ++i) {
        const int K = 2 * ki + 1, u = w + PS_WAVES * i;
        t.off[ki][w][i] = ((u / K) * 16 * PS_TP + (u % K)) * 4;
      }
  return t;
}
__constant__ PsUnitTab ps_unit_tab = ps_make_unit_tab();

__device__ __forceinline__ void ps_load_weights(const PS_G float* w, int n_u, int wave, int lane, f32x4 (&a)[PS_MAXU]) {
  // units beyond n_u repeat the last one (an L1 hit; never used: the MFMA loop tests u < n_u).  The unit index is wave-uniform:
  // clamp and scale on the scalar unit, one 32-bit lane offset for all eight loads.
  const PS_G char* base = (const PS_G char*)w;
  const unsigned lane_off = (unsigned)lane * 16u;
#pragma unroll
  for (int i = 0; i < PS_MAXU; ++i) {
    const int u = wave + PS_WAVES * i;
    const unsigned uo = (unsigned)__builtin_amdgcn_readfirstlane(u < n_u ? u : n_u - 1) * 1024u;
    a[i] = *(const PS_G f32x4*)(base + (uo + lane_off));
  }
}
// epilogue operand indices of thread tid relative to the record's bias / vector pointers
__device__ __forceinline__ void ps_epi_idx(int kf, int rows_left, int gate_H, int tid, int blk, int& i0, int& i1) {
  if (kf & PF_GATE) { i0 = blk * 8 + (tid & 7); i1 = gate_H + i0; }   // [8 tanh | 8 sigmoid] rows of channels 8 * mb + ch
  else {
    i0 = blk * 16 + (tid & 15);
    i0 = i0 < rows_left ? i0 : rows_left - 1;
    i1 = i0;
  }
}
// column / attention / merge steps: packed parameters (row tid of the pack; attention: 512 rows) and the per-item vector
__device__ __forceinline__ void ps_prefetch_light(const ps_i4& r, int tid, PsPre& pre) {
  const int kind = PR_I(r, 0) & 0xff;
  const PS_G f32x4* pk = PR_P(const f32x4, r, 9);
  const int row = tid & (kind == PK_ATT ? 511 : 255);
  pre.pk0 = pk[row * 2];
  pre.pk1 = pk[row * 2 + 1];
  const int C = PR_I(r, 1) & 0xffff;
  int i0 = tid & 255; i0 = i0 < C ? i0 : C - 1; i0 = i0 < 0 ? 0 : i0;
  pre.ec0 = PR_P(const float, r, 8)[i0];
}
// matrix steps: weight fragments of the first row block and the epilogue vectors (straight-line: behind a branch the loads cost a
// conservative wait in the epilogue, measured +0.5 .. 1.3 k cycles per step)
__device__ __forceinline__ void ps_prefetch(const ps_i4& r, int tid, int wave, int lane, PsPre& pre) {
  const int kf = PR_I(r, 0);
  int i0, i1;
  ps_epi_idx(kf, PR_B(r, 3), PR_B(r, 7), tid, 0, i0, i1);
  const PS_G float* b = PR_P(const float, r, 7);
  const PS_G float* c = PR_P(const float, r, 8);
  pre.eb0 = b[i0]; pre.eb1 = b[i1]; pre.ec0 = c[i0]; pre.ec1 = c[i1];
  ps_load_weights(PR_P(const float, r, 6), PR_B(r, 0) & 0x7f, wave, lane, pre.a);
}

// LDS-DMA load of one dword per lane into a scratch slot of the LDS: a load WITHOUT a register destination.  Used to pull lines into
// this XCD's L2 ahead of time.  The compiler does not see it (inline asm): it is absent from its s_waitcnt bookkeeping, which is
// what is wanted here -- nothing ever waits for it -- and harmless for the loads the compiler does count (vmcnt retires in order:
// a counted wait can only wait longer, never shorter).  M0 holds the LDS destination and is restored (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void ps_glds_dword(const PS_G char* g, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
// the weight fragments wave `wave` will stream in a later matrix step (units u = wave + 8 i of nblk row blocks, 1 KiB each): one dword
// per 64 bytes, four units per instruction
__device__ __forceinline__ void ps_touch_weights(const PS_G char* w, int n_u, int nblk, unsigned stride_bytes, int wave, int lane, unsigned lds_dst) {
  const unsigned lo = (unsigned)(lane & 15) * 64u;
  for (int blk = 0; blk < nblk; ++blk) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int u = wave + PS_WAVES * (h * 4 + (lane >> 4));
      u = u < n_u ? u : n_u - 1;  // (beyond the last unit: a line that is fetched anyway)
      ps_glds_dword(w + ((unsigned)u * 1024u + lo), lds_dst);
    }
    w += stride_bytes;
  }
}

// Placed where a poll has just completed: the record of step s + 1 (requested at the top of this step, older than every poll load) is
// certainly there -- touching it HERE makes the compiler account for its wait at a point where it costs nothing.  Without this the
// hand-over rv = rvB at the loop's back edge is the first use, and because loads and stores share the in-order vmcnt counter the
// compiler waits there for vmcnt(0): for the step's own output stores (an sc1 write round trip per step).
#define PS_REC_READY() asm volatile("" : "+v"(rvB))
// cycle stamps for tools/ps_trace.py: only in the -DPS_TRACE build (libvits_mi355_pstrace.so) -- at 2 waves per SIMD every instruction
// of a step is on its critical path, and eight guarded stamps per step were ~40 of them
#ifdef PS_TRACE
#define PS_STAMP(k) do { if (call.trace && tid0 == 0) call.trace[((long long)rank * PS_MAX_STEPS + s) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PS_STAMP(k) do { } while (0)
#endif

// LDS of the kernel (floats).  Matrix steps: operand window + partial tiles + spline scratch; attention blocks: Q / K / V tiles and
// the two relative-position tables alias the operand window, scores / probabilities alias the partial tiles.
#define PS_LDS_TILE (3 * 16 * (PS_DKP + 4) + (10 + 12) * (PS_DKP + 4))   // >= PS_MAXC * PS_TP
#define PS_LDS_MRED (PS_WAVES * 256)
static_assert(PS_LDS_TILE >= PS_MAXC * PS_TP, "operand window does not fit");

__global__ void __launch_bounds__(PS_THREADS) persist_kernel(const PProgram* __restrict__ prog, const PCall call) {
  __shared__ __attribute__((aligned(16))) float tile[PS_LDS_TILE];      // MFMA operand window [C_in][16 + taps - 1] (pitch 21) / attention tiles
  __shared__ __attribute__((aligned(16))) float mred[PS_LDS_MRED];      // partial tiles of the 8 waves / attention scores
  __shared__ float hb[32 * 16];            // ConvFlow.proj output of the tile (spline parameters)
  __shared__ float xs[3 * PS_MAXC];        // column steps: x_in at t - d, t, t + d
  __shared__ float red[4 * 16];            // block reductions (one 16-float scratch per call site)
  __shared__ float dma_sink[64];           // destination of the L2 pre-touch loads (never read)
  __shared__ __attribute__((aligned(16))) float att_r[PS_WAVES * 256];  // attention blocks: partial q E_k^T tiles of the 8 waves
  __shared__ float att_p[16 * 17];         // attention blocks: probabilities [query][key], pitch 17 (conflict-free MFMA A reads)
  const int tid0 = threadIdx.x;
  const int wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int rank = blockIdx.x, P = gridDim.x;
  const int n_steps = prog->n_steps, seg_step = prog->seg_step;
  int T = prog->T, Tp = prog->Tp;
  const PS_G PRec* recs = (const PS_G PRec*)prog->recs;
  // record of step s for this worker: lanes 0..7 read its eight dwordx4 (the other lanes read along: same lines)
  auto load_rec = [&](int s) -> ps_i4 {
    const PS_G ps_i4* p = (const PS_G ps_i4*)(recs + ((size_t)(s < n_steps ? s : n_steps - 1) * P + rank));
    return p[tid0 & 7];
  };
  ps_i4 rvB = load_rec(0);  // the record of the NEXT step: requested at the top of a step, touched right after that step's poll
  PsCtx cx;
  {
    unsigned e = __hip_atomic_load(&call.ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    cx.epoch = __builtin_amdgcn_readfirstlane(e ? e : 1u);  // 0 marks "never written"
  }
  cx.aborted = 0; cx.spins = 0; cx.ctl = call.ctl;
  {
    const int lim = __builtin_amdgcn_readfirstlane(__hip_atomic_load(call.dbg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    cx.limit = lim > 0 ? lim : PS_SPIN_LIMIT;
  }
  const unsigned epoch = cx.epoch;
  int len_raw;
  {
    int zero = 0;
    asm volatile("" : "+v"(zero));
    len_raw = ((const PS_G int*)prog->len)[zero];  // vector load (stays off the scalar counter)
  }
  // Loop-invariant uniform values go into SGPRs (readfirstlane): as VGPRs they are part of the block of loop-carried registers the
  // compiler shuffles at the step loop's back edge, right behind the step's output stores -- and overwriting a register that was a
  // store's data operand costs an s_waitcnt vmcnt(0) there, i.e. the completion of the step's sc1 stores, every step.
  auto uni = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
  len_raw = __builtin_amdgcn_readfirstlane(len_raw);
  const float ea_m = uni(((const PS_G float*)prog->ea_m)[0]), ea_is = uni(expf(-((const PS_G float*)prog->ea_logs)[0]));  // used by the very last epilogue
  const float spline_cst = uni(logf(expf(1.f - 1e-3f) - 1.f));  // boundary derivative parameter (transforms.py:100-103); once, not per column
  PsPre pre;

  for (int s = 0; s < n_steps; ++s) {
    // (opaque per step: every per-thread index below derives from this copy, so that the compiler does not hoist the address
    //  arithmetic of ALL phases out of the step loop and spill it)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    // (the same for the wave index: hipcc hoists every wave-derived scalar expression of every step kind out of the step loop --
    //  ~140 of them -- spills them to VGPR lanes in the prologue and reads them back with v_readlane where one SALU instruction would do)
    int wave = wave0;
    asm volatile("" : "+s"(wave));
    const ps_i4 rv = rvB;
    const int kf = PR_I(rv, 0), kind = kf & 0xff;
    rvB = load_rec(s + 1);  // (an L2 hit; older than everything else this step requests)
    if (s == seg_step) {
      // second half of a two-halves program: from here on the frame-side geometry, and the frame count this launch's PK_DUR step
      // publishes.  EVERY worker passes here (also the ones that idle in this step), so it sits in front of the idle test; a worker
      // that has been idle through the whole duration predictor waits here for ~100 us -- ps_again backs off (s_sleep) after a few rounds.
      const PS_G ll_t* lc = (const PS_G ll_t*)prog->len2;
      ll_t q;
      bool pending;
      do {
        q = ll_load(lc);
        pending = PS_PENDING(ll_bad(q, epoch));
      } while (ps_again(cx, pending));
      len_raw = __builtin_amdgcn_readfirstlane((int)(unsigned)q);  // (the cell's value field holds the integer)
      T = prog->T2; Tp = prog->Tp2;
    }
    if (kind == PK_IDLE) continue;  // nothing to do and nothing to wait for in this step
    PS_STAMP(0);
    // This step's own operands (weight fragments, packed parameters, epilogue vectors), requested FIRST: the record decode and the
    // poll set-up below take about as long as they need to arrive, and every poll load is younger than they are.  (They used to be
    // requested one step ahead, after the previous step's MFMAs: as loop-carried registers written on five different paths they
    // cost an s_waitcnt vmcnt(0) plus 40 moves at the end of every step -- the whole fetch latency, exposed: 2 k of a 12 k-cycle step.)
    // Two request sequences, each followed by its kinds' whole bodies (no join before the end of the step: a join right behind
    // conditional loads costs a conservative wait): matrix steps request weight fragments + epilogue vectors + packs, the column /
    // attention kinds only their packed parameters and per-item vector (three loads instead of fourteen).
    const int L = len_raw < T ? len_raw : T;
    if (kind != PK_MM) {
    ps_prefetch_light(rv, tid, pre);
    __syncthreads();  // the previous step's readers of the LDS buffers are done

    if (kind == PK_DDS) {
      // ================================================================== DDSConv column step
      const int D = PR_I(rv, 1) & 0xffff, d = PR_I(rv, 1) >> 16, t = PR_I(rv, 2);
      const bool dw = (kf & PF_DW) != 0;
      const float invD = 1.0f / (float)D;
      const PS_G ll_t* xin = PR_P(const ll_t, rv, 0);
      const PS_G ll_t* y2 = (kf & PF_FIN_LN) ? PR_P(const ll_t, rv, 1) : nullptr;
      const PS_G ll_t* zc = (kf & PF_FIN_PRE) ? PR_P(const ll_t, rv, 2) : nullptr;
      PS_G ll_t* xout = PR_P(ll_t, rv, 3);
      PS_G ll_t* bout = PR_P(ll_t, rv, 4);
      const int c = (wave & 3) * 64 + lane, h = wave >> 2;  // (== tid & 255, tid >> 8; the half is wave-uniform: scalar conditions)
      const bool cok = c < D;
      {
        if (t >= L) {  // padding column (worker-uniform): zeros, nothing to wait for
          if (h == 1 && cok) {
            ll_store(xout + (long long)t * D + c, 0.f, epoch);
            if (dw) ll_store(bout + (long long)t * D + c, 0.f, epoch);
          }
          PS_STAMP(3);
          continue;
        }
        // slots of this thread: half 1 -> column t; half 0 -> columns t - d and t + d (depthwise steps only)
        const int t0 = h ? t : t - d, t1 = t + d;
        const bool n0 = cok && (h ? true : (dw && t0 >= 0)), n1 = cok && !h && dw && t1 < L;
        const int tc0 = t0 < 0 ? 0 : t0, tc1 = t1 < Tp ? t1 : Tp - 1;
        float x0 = 0.f, y0 = 0.f, z0 = 0.f, x1 = 0.f, y1 = 0.f, z1 = 0.f;
        {
          unsigned o0 = (unsigned)(tc0 * D + c) * 8u, o1 = (unsigned)(tc1 * D + c) * 8u, oz0 = (unsigned)tc0 * 8u, oz1 = (unsigned)tc1 * 8u;
          bool pending;
          do {
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(oz0), "+v"(oz1));  // (addresses stay inside the loop body: no hoisted 64-bit pairs)
            ll_t qx0 = 0, qy0 = 0, qz0 = 0, qx1 = 0, qy1 = 0, qz1 = 0;
            if (n0) {
              qx0 = ll_load_off(xin, o0);
              if (y2) qy0 = ll_load_off(y2, o0);
              if (zc) qz0 = ll_load_off(zc, oz0);
            }
            if (n1) {
              qx1 = ll_load_off(xin, o1);
              if (y2) qy1 = ll_load_off(y2, o1);
              if (zc) qz1 = ll_load_off(zc, oz1);
            }
            unsigned bad = 0;
            if (n0) bad |= ll_bad(qx0, epoch) | (y2 ? ll_bad(qy0, epoch) : 0u) | (zc ? ll_bad(qz0, epoch) : 0u);
            if (n1) bad |= ll_bad(qx1, epoch) | (y2 ? ll_bad(qy1, epoch) : 0u) | (zc ? ll_bad(qz1, epoch) : 0u);
            x0 = ll_val(qx0); y0 = ll_val(qy0); z0 = ll_val(qz0); x1 = ll_val(qx1); y1 = ll_val(qy1); z1 = ll_val(qz1);
            pending = PS_PENDING(bad);
          } while (ps_again(cx, pending));
        }
        PS_STAMP(1); PS_REC_READY();
        const float par[8] = {pre.pk0[0], pre.pk0[1], pre.pk0[2], pre.pk0[3], pre.pk1[0], pre.pk1[1], pre.pk1[2], pre.pk1[3]};
        if (zc) { x0 = par[0] * z0 + par[1] + x0; x1 = par[0] * z1 + par[1] + x1; }  // ConvFlow.pre(x0) + g  (modules.py:365-366)
        if (y2) {  // x + gelu(LN2(y2)), two-pass statistics like F.layer_norm; both slots of a half in the same reductions
          float m0 = n0 ? y0 : 0.f, m1 = n1 ? y1 : 0.f;
          ps_half_sum2(m0, m1, red, wave, lane);
          m0 *= invD; m1 *= invD;
          const float e0 = y0 - m0, e1 = y1 - m1;
          float q0 = n0 ? e0 * e0 : 0.f, q1 = n1 ? e1 * e1 : 0.f;
          ps_half_sum2(q0, q1, red + 16, wave, lane);
          x0 += PS_GELU(e0 * (1.0f / sqrtf(q0 * invD + 1e-5f)) * par[0] + par[1]);
          x1 += PS_GELU(e1 * (1.0f / sqrtf(q1 * invD + 1e-5f)) * par[0] + par[1]);
        }
        // x = (x + y) * mask: columns outside [0, L) are zero (never polled)
        const float xi0 = (n0 && t0 < L) ? x0 : 0.f, xi1 = n1 ? x1 : 0.f;
        if (cok) {
          if (h) { xs[PS_MAXC + c] = xi0; ll_store(xout + (long long)t * D + c, xi0, epoch); }
          else { xs[c] = xi0; xs[2 * PS_MAXC + c] = xi1; }
        }
        if (dw) {
          __syncthreads();
          // depthwise conv (modules.py:100), LN1, GELU: the 256 threads of half 1 (one erf so far; half 0 had two and runs
          // along for the barriers)
          float v = 0.f;
          if (cok) v = par[2] + par[3] * xs[c] + par[4] * xs[PS_MAXC + c] + par[5] * xs[2 * PS_MAXC + c];
          float m = (cok && h) ? v : 0.f, dummy = 0.f;
          ps_half_sum2(m, dummy, red + 32, wave, lane);
          m *= invD;
          const float e = v - m;
          float q = (cok && h) ? e * e : 0.f;
          ps_half_sum2(q, dummy, red + 48, wave, lane);
          if (cok && h) ll_store(bout + (long long)t * D + c, PS_GELU(e * (1.0f / sqrtf(q * invD + 1e-5f)) * par[6] + par[7]), epoch);
        }
      }
      PS_STAMP(3);
      continue;
    }

    if (kind == PK_MERGE) {
      // ================================================================== attention merge, column t:
      // out = sum_b w_b O_b / sum_b w_b l_b, w_b = e^(m_b - max m) over the key tiles b; thread = (d = tid & 127, head = tid >> 7)
      const int C = PR_I(rv, 1), t = PR_I(rv, 2), dk = PR_I(rv, 3) & 0xff, nh = PR_I(rv, 3) >> 8, dk2 = dk + 2;
      const PS_G ll_t* ap = PR_P(const ll_t, rv, 0);
      PS_G ll_t* out = PR_P(ll_t, rv, 3);
      const int d = (wave & 1) * 64 + lane, hd = wave >> 1;  // (== tid & 127, tid >> 7; the head is wave-uniform)
      const bool ok = d < dk && hd < nh;
      const int nkt = (L + 15) >> 4;
      const long long kstr = PR_B(rv, 0);
      {
        float r = 0.f;
        if (t < L) {
          float M = -3.0e38f, num = 0.f, den = 0.f;
          for (int k0 = 0; k0 < nkt; k0 += PS_MKT) {  // PS_MKT key tiles per poll round: one round trip for T <= 16 PS_MKT
            float ov[PS_MKT], mv[PS_MKT], lv[PS_MKT];
            unsigned ob = (unsigned)((t * nh + (hd < nh ? hd : 0)) * dk2) * 8u, od = (unsigned)(d < dk ? d : 0) * 8u;
            bool pending;
            do {
              asm volatile("" : "+v"(ob), "+v"(od));
              ll_t qo[PS_MKT], qm[PS_MKT], ql[PS_MKT];
#pragma unroll
              for (int k = 0; k < PS_MKT; ++k) { qo[k] = 0; qm[k] = 0; ql[k] = 0; }
              if (ok) {
#pragma unroll
                for (int k = 0; k < PS_MKT; ++k)
                  if (k0 + k < nkt) {
                    const PS_G ll_t* b = ap + (k0 + k) * kstr;
                    qo[k] = ll_load_off(b, ob + od);
                    qm[k] = ll_load_off(b, ob + (unsigned)dk * 8u);
                    ql[k] = ll_load_off(b, ob + (unsigned)(dk + 1) * 8u);
                  }
              }
              unsigned bad = 0;
#pragma unroll
              for (int k = 0; k < PS_MKT; ++k) {
                if (ok && k0 + k < nkt) bad |= ll_bad(qo[k], epoch) | ll_bad(qm[k], epoch) | ll_bad(ql[k], epoch);
                ov[k] = ll_val(qo[k]); mv[k] = ll_val(qm[k]); lv[k] = ll_val(ql[k]);
              }
              pending = PS_PENDING(bad);
            } while (ps_again(cx, pending));
#pragma unroll
            for (int k = 0; k < PS_MKT; ++k)
              if (k0 + k < nkt) {
                const float Mn = fmaxf(M, mv[k]);
                const float a0 = __expf(M - Mn), a1 = __expf(mv[k] - Mn);
                num = num * a0 + ov[k] * a1;
                den = den * a0 + lv[k] * a1;
                M = Mn;
              }
          }
          r = den > 0.f ? num / den : 0.f;
        }
        PS_STAMP(1); PS_REC_READY();
        if (ok) ll_store(out + (long long)t * C + hd * dk + d, r, epoch);
      }
      PS_STAMP(3);
      continue;
    }

    if (kind == PK_DUR) {
      // ================================================================== durations, cumulative sums, frame count (one worker)
      // w = ceil(exp(logw) * length_scale) * mask, cumsum, y_length = max(sum, 1)   (models.py:1689-1694; durations_kernel); thread = token
      const int Tx = PR_I(rv, 1), Tcap = PR_I(rv, 2);
      const PS_G ll_t* lwc = PR_P(const ll_t, rv, 0);
      int* redi = reinterpret_cast<int*>(red);
      float length_scale = call.length_scale;
      if (call.dv) length_scale = call.dv->scales[1];
      const bool tok = tid < Tx;
      float lw = 0.f;
      if (!(kf & PF_PLAIN_IN)) {  // wait for the duration predictor of THIS launch (also when the durations are pinned: the benchmark's
        // dependency chain is the production one)
        unsigned o = (unsigned)(tok ? tid : 0) * 8u;
        bool pending;
        do {
          asm volatile("" : "+v"(o));
          ll_t q = (ll_t)epoch << 32;
          if (tok) q = ll_load_off(lwc, o);
          lw = ll_val(q);
          pending = PS_PENDING(ll_bad(q, epoch));
        } while (ps_again(cx, pending));
      }
      PS_STAMP(1); PS_REC_READY();
      int d = 0;
      if (tok && tid < L) d = call.forced ? call.forced[tid] : (int)ceilf(expf(lw) * length_scale);
      if (d < 0) d = 0;
      // inclusive scan: DPP inside a wave, wave totals through LDS
      int v = d;
#define PS_IDPP(x, CTRL, RM) __builtin_amdgcn_update_dpp(0, (x), (CTRL), (RM), 0xF, false)
      v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
      v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
      v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
      v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
      v += PS_IDPP(v, 0x142, 0xA);                                    // row_bcast:15 -> rows 1 and 3
      v += PS_IDPP(v, 0x143, 0xC);                                    // row_bcast:31 -> rows 2 and 3
#undef PS_IDPP
      if (lane == 63) redi[wave] = v;
      __syncthreads();
      int base = 0, total = 0;
#pragma unroll
      for (int w = 0; w < PS_WAVES; ++w) { const int x = redi[w]; total += x; base += w < wave ? x : 0; }
      const int cum = base + v;
      int yl = total < 1 ? 1 : total;
      if (Tcap > 0 && yl > Tcap) { if (tid == 0) atomicOr((int*)prog->err, 4); yl = Tcap; }
      if (tok) {
        PR_P(int, rv, 1)[tid] = d;
        PR_P(int, rv, 2)[tid] = cum;
        ll_store(PR_P(ll_t, rv, 3) + tid, __int_as_float(cum), epoch);
      }
      if (tid == 0) {
        PR_P(int, rv, 4)[0] = yl;
        if (PR_P(long long, rv, 5)) PR_P(long long, rv, 5)[0] = yl;
        ll_store(PR_P(ll_t, rv, 6), __int_as_float(yl), epoch);
      }
      PS_STAMP(3);
      continue;
    }

    if (kind == PK_EXPAND) {
      // ================================================================== length regulator + prior sample, frame f:
      // z_p[:, f] = m_p[:, tok] + eps * exp(logs_p[:, tok]) * noise_scale, tok = the token whose span holds f   (models.py:1696-1700 as a
      // gather, expand_prior_kernel); thread = channel
      const int I = PR_I(rv, 1), f = PR_I(rv, 2), Tx = PR_I(rv, 3), pT = PR_B(rv, 4);
      const bool plain = (kf & PF_PLAIN_IN) != 0;
      PS_G ll_t* out = PR_P(ll_t, rv, 3);
      PS_G float* oplain = PR_P(float, rv, 4);
      int* redi = reinterpret_cast<int*>(red);
      const int c = (wave & 3) * 64 + lane, h = wave >> 2;
      const bool cok = c < I && h == 0;
      float o = 0.f;
      if (f < L) {  // (worker-uniform) frames beyond the utterance: zeros, nothing to wait for
        // tok = number of tokens whose cumulative duration is <= f (= first j with cum[j] > f; f < y_length <= cum[Tx - 1])
        const bool iok = tid < Tx;
        int cv = 0;
        if (plain) { if (iok) cv = PR_P(const int, rv, 1)[tid]; }
        else {
          const PS_G ll_t* cc = PR_P(const ll_t, rv, 1);
          unsigned oc = (unsigned)(iok ? tid : 0) * 8u;
          bool pending;
          do {
            asm volatile("" : "+v"(oc));
            ll_t q = (ll_t)epoch << 32;
            if (iok) q = ll_load_off(cc, oc);
            cv = (int)(unsigned)q;
            pending = PS_PENDING(ll_bad(q, epoch));
          } while (ps_again(cx, pending));
        }
        const int cnt = __builtin_popcountll(__builtin_amdgcn_ballot_w64(iok && cv <= f));
        if (lane == 0) redi[wave] = cnt;
        __syncthreads();
        int tk = 0;
#pragma unroll
        for (int w = 0; w < PS_WAVES; ++w) tk += redi[w];
        tk = tk < Tx ? tk : Tx - 1;
        float mu = 0.f, ls = 0.f;
        if (plain) {
          const PS_G float* st = PR_P(const float, rv, 0);
          if (cok) { mu = st[c * Tx + tk]; ls = st[(I + c) * Tx + tk]; }
        } else {
          const PS_G ll_t* st = PR_P(const ll_t, rv, 0);
          unsigned om = (unsigned)(tk * 2 * I + (cok ? c : 0)) * 8u;
          bool pending;
          do {
            asm volatile("" : "+v"(om));
            ll_t q0 = (ll_t)epoch << 32, q1 = q0;
            if (cok) { q0 = ll_load_off(st, om); q1 = ll_load_off(st, om + (unsigned)I * 8u); }
            mu = ll_val(q0); ls = ll_val(q1);
            pending = PS_PENDING(ll_bad(q0, epoch) | ll_bad(q1, epoch));
          } while (ps_again(cx, pending));
        }
        PS_STAMP(1);
        float noise_scale = call.noise_scale;
        unsigned long long seed = call.seed;
        if (call.dv) { noise_scale = call.dv->scales[0]; seed = call.dv->seed; }
        if (call.solo && call.item_seeds) seed = call.item_seeds[0];
        if (cok) {
          const float e = call.noise_prior ? call.noise_prior[(long long)c * call.noise_stride + f] : philox_normal(seed, 2, (uint32_t)c, (uint32_t)f);
          o = mu + e * expf(ls) * noise_scale;
        }
      }
      PS_REC_READY();
      if (cok) {
        if (out) ll_store(out + (long long)f * I + c, o, epoch);
        if (oplain && f < pT) oplain[(long long)c * pT + f] = o;
      }
      PS_STAMP(3);
      continue;
    }

    if (kind == PK_LN || kind == PK_EMB || kind == PK_COUPLE) {
      // ================================================================== LayerNorm / embedding / coupling tail, column t
      const int C = PR_I(rv, 1) & 0xffff, t = PR_I(rv, 2), pT = PR_B(rv, 4);
      PS_G ll_t* out = PR_P(ll_t, rv, 3);
      PS_G float* oplain = PR_P(float, rv, 4);
#define par_g pre.pk0[0]
#define par_b pre.pk0[1]
#define par_bias pre.pk0[2]
#define par_vec pre.ec0
      const int c = (wave & 3) * 64 + lane, h = wave >> 2;  // (== tid & 255, tid >> 8)
      const bool cok = c < C && h == 0;  // one column per worker: the channel threads of half 0
      {
        float o = 0.f;
        if (t < L) {  // (worker-uniform) padding columns: zeros, nothing to wait for
          if (kind == PK_EMB) {
            // x = emb[id] * sqrt(H) (+ the speaker vector when the first layer is the conditioned one)   (models.py:318-322)
            long long id = call.ids[t];
            if (id < 0 || id >= PR_B(rv, 1)) { if (tid == 0) atomicOr((int*)prog->err, 1); id = 0; }
            if (cok) o = PR_P(const float, rv, 0)[id * C + c] * __int_as_float(PR_B(rv, 0)) + par_vec;
          } else if (kind == PK_LN) {
            const int np = PR_I(rv, 3);
            const PS_G ll_t* part = PR_P(const ll_t, rv, 0);
            const PS_G ll_t* res = PR_P(const ll_t, rv, 1);
            const PS_G ll_t* base = PR_P(const ll_t, rv, 2);
            const long long pstr = ((long long)(unsigned)PR_B(rv, 1) << 32) | (unsigned)PR_B(rv, 0);
            float v = 0.f, bs = 0.f;
            {
              unsigned o0 = (unsigned)(t * C + c) * 8u;
              bool pending;
              do {
                asm volatile("" : "+v"(o0));
                ll_t q[4] = {0, 0, 0, 0}, qr = 0, qb = 0;
                if (cok) {
#pragma unroll
                  for (int k = 0; k < 4; ++k)
                    if (k < np) q[k] = ll_load_off(part + k * pstr, o0);
                  if (res) qr = ll_load_off(res, o0);
                  if (base) qb = ll_load_off(base, o0);
                }
                unsigned bad = 0;
                v = 0.f;
                if (cok) {
#pragma unroll
                  for (int k = 0; k < 4; ++k)
                    if (k < np) { bad |= ll_bad(q[k], epoch); v += ll_val(q[k]); }
                  if (res) { bad |= ll_bad(qr, epoch); v += ll_val(qr); }
                  if (base) bad |= ll_bad(qb, epoch);
                }
                bs = ll_val(qb);
                pending = PS_PENDING(bad);
              } while (ps_again(cx, pending));
            }
            PS_STAMP(1); PS_REC_READY();
            v += par_bias;  // (after the poll: the packed parameters were requested at the top of the step)
            if (kf & PF_LN) {
              const float invC = 1.0f / (float)C;
              float m = cok ? v : 0.f, dummy = 0.f;
              ps_half_sum2(m, dummy, red, wave, lane);
              m *= invC;
              const float e = v - m;
              float q = cok ? e * e : 0.f;
              ps_half_sum2(q, dummy, red + 16, wave, lane);
              o = e * (1.0f / sqrtf(q * invC + 1e-5f)) * par_g + par_b + par_vec + bs;
            } else {
              o = v + bs;
            }
          } else {
            // new z = cat(x0, (x1 - m) * mask) with the following Flip folded in (models.py:390-392, modules.py:270-277)
            const int np = PR_I(rv, 3) & 0xffff, H = PR_I(rv, 3) >> 16;
            const PS_G ll_t* part = PR_P(const ll_t, rv, 0);
            const PS_G ll_t* u = PR_P(const ll_t, rv, 1);
            const PS_G float* up = (kf & PF_PLAIN_IN) ? PR_P(const float, rv, 2) : nullptr;
            const long long pstr = ((long long)(unsigned)PR_B(rv, 1) << 32) | (unsigned)PR_B(rv, 0);
            const int r = c < H ? c : c - H;               // c < H: copy of x0 ; else row r of the transformed half
            const int src = c < H ? 2 * H - 1 - c : H - 1 - r;
            float uv = 0.f, mv = 0.f;
            if (up) { if (cok) uv = up[(long long)src * pT + t]; }
            {
              unsigned ou = (unsigned)(t * 2 * H + src) * 8u, om = (unsigned)(t * H + r) * 8u;
              const bool needm = cok && c >= H;
              bool pending;
              do {
                asm volatile("" : "+v"(ou), "+v"(om));
                ll_t qu = 0, q[4] = {0, 0, 0, 0};
                if (cok && !up) qu = ll_load_off(u, ou);
                if (needm) {
#pragma unroll
                  for (int k = 0; k < 4; ++k)
                    if (k < np) q[k] = ll_load_off(part + k * pstr, om);
                }
                unsigned bad = 0;
                if (cok && !up) { bad |= ll_bad(qu, epoch); uv = ll_val(qu); }
                mv = 0.f;
                if (needm) {
#pragma unroll
                  for (int k = 0; k < 4; ++k)
                    if (k < np) { bad |= ll_bad(q[k], epoch); mv += ll_val(q[k]); }
                }
                pending = PS_PENDING(bad);
              } while (ps_again(cx, pending));
            }
            PS_STAMP(1); PS_REC_READY();
            o = c < H ? uv : uv - (mv + par_bias);
          }
        }
        if (cok) {
          if (out) ll_store(out + (long long)t * C + c, o, epoch);
          if (oplain && t < pT) oplain[(long long)c * pT + t] = o;
        }
      }
      PS_STAMP(3);
      continue;
    }
#undef par_g
#undef par_b
#undef par_bias
#undef par_vec

    if (kind == PK_ATT) {
      // ================================================================== attention block (head, query tile, key tile)
      // s_ij = q~_i . k_j (+ q~_i . E_k[j - i + W] inside the band) ; p_ij = exp(s_ij - m_i) ; O_i = sum_j p_ij (v_j + E_v[j - i + W])
      // (attentions.py:165-260 in the exact banded form of relpos_attention_kernel); partial (O, m, l) per block, merged by PK_MERGE
      const int dk = PR_I(rv, 1) & 0xff, nh = (PR_I(rv, 1) >> 8) & 0xff, W = PR_I(rv, 1) >> 16, dk2 = dk + 2, H3 = 3 * nh * dk;
      const int i0 = PR_I(rv, 2) & 0xffff, j0 = PR_I(rv, 2) >> 16, hd = PR_I(rv, 3) & 0xff, kt = PR_I(rv, 3) >> 8;
      const PS_G ll_t* qkv = PR_P(const ll_t, rv, 0);
      PS_G ll_t* ap = PR_P(ll_t, rv, 1);
      // rows of pitch dk + 4 floats: 16-byte aligned (the dot products read float4) and, at 16 bytes per lane, conflict-free for
      // the 16 rows a lane group touches ((dk + 4) / 4 is odd for dk = 8 (2 n + 1) - 4 ... checked for dk = 96: 25)
      float* Qs = tile;                  // [16][dk + 4]
      float* Ks = Qs + 16 * (PS_DKP + 4);
      float* Vs = Ks + 16 * (PS_DKP + 4);
      float* Ek = Vs + 16 * (PS_DKP + 4);  // [2W + 1][dk + 4] (+ row 9 zeroed)
      float* Ev = Ek + 10 * (PS_DKP + 4);  // [12][dk + 4]: rows 2W + 1 .. 11 zero (the band product runs over 12 relative positions)
      float* Sp = mred;                  // [8 waves][4][64] partial q k^T tiles (MFMA accumulator layout)
      float* Rp = att_r;                 // the same for q E_k^T
      float* Ps = att_p;                 // [16][17] probabilities
      const int dkp = dk + 4;
      const float scale = 1.0f / sqrtf((float)dk);
      {
        if (i0 >= L || j0 >= L) { PS_STAMP(3); continue; }  // nothing to compute: the merge step never looks at these blocks
        // ---- gather q (16 x dk), k, v tiles: thread = (d = tid & 127, rows (tid >> 7) + 4 k)
        {
          const int d = (wave & 1) * 64 + lane, r0 = wave >> 1;  // (== tid & 127, tid >> 7; the row group is wave-uniform)
          const bool dok = d < dk;
          unsigned oq[4], okv[4];
          bool nq[4], nk[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = r0 + 4 * k;
            nq[k] = dok && i0 + r < L;
            nk[k] = dok && j0 + r < L;
            const int tq = i0 + r < Tp ? i0 + r : Tp - 1, tk = j0 + r < Tp ? j0 + r : Tp - 1;
            oq[k] = (unsigned)(tq * H3 + hd * dk + (dok ? d : 0)) * 8u;
            okv[k] = (unsigned)(tk * H3 + nh * dk + hd * dk + (dok ? d : 0)) * 8u;
          }
          float vq[4], vk[4], vv[4];
          bool pending;
          do {
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(oq[k]), "+v"(okv[k]));
            ll_t qq[4] = {0, 0, 0, 0}, qk[4] = {0, 0, 0, 0}, qv[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (nq[k]) qq[k] = ll_load_off(qkv, oq[k]);
              if (nk[k]) { qk[k] = ll_load_off(qkv, okv[k]); qv[k] = ll_load_off(qkv, okv[k] + (unsigned)(nh * dk) * 8u); }
            }
            unsigned bad = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (nq[k]) bad |= ll_bad(qq[k], epoch);
              if (nk[k]) bad |= ll_bad(qk[k], epoch) | ll_bad(qv[k], epoch);
              vq[k] = ll_val(qq[k]); vk[k] = ll_val(qk[k]); vv[k] = ll_val(qv[k]);
            }
            pending = PS_PENDING(bad);
          } while (ps_again(cx, pending));
          PS_STAMP(1); PS_REC_READY();
          if (W > 0) {  // E_k[tid], E_k[tid + 512], E_v[tid], E_v[tid + 512] of the flat [2W + 1][dk] tables -> rows of pitch dk + 1:
            // the score pass reads row (j - i + W) per LANE; at pitch dk = 96 all nine rows of a 16-lane group fell on one bank
            const int tab = (2 * W + 1) * dk;
            const int ra = tid / dk, rb = (tid + 512) / dk;
            if (tid < tab) { Ek[ra * dkp + (tid - ra * dk)] = pre.pk0[0]; Ev[ra * dkp + (tid - ra * dk)] = pre.pk0[2]; }
            if (tid + 512 < tab) { Ek[rb * dkp + (tid + 512 - rb * dk)] = pre.pk0[1]; Ev[rb * dkp + (tid + 512 - rb * dk)] = pre.pk0[3]; }
            if (tid < dkp) Ek[9 * dkp + tid] = 0.f;
            if (tid < 3 * dkp && 2 * W + 1 <= 9) Ev[(9 * dkp) + tid] = 0.f;                                   // rows 9 .. 11
            for (int r = 2 * W + 1; r < 9; ++r) if (tid < dkp) Ev[r * dkp + tid] = 0.f;                        // (W < 4)
          }
          if (dok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = r0 + 4 * k;
              Qs[r * dkp + d] = nq[k] ? vq[k] * scale : 0.f;
              Ks[r * dkp + d] = nk[k] ? vk[k] : 0.f;
              Vs[r * dkp + d] = nk[k] ? vv[k] : 0.f;
            }
          }
        }
        __syncthreads();
        PS_STAMP(2);
        // ---- scores on the matrix cores (round 4; the float4 dot products this replaces were 2.2 k + 3.9 k cycles of VALU and LDS
        // issue per block).  S = Q K^T and, for blocks that touch the band, R = Q E_k^T (R[i][r] = q_i . E_k[r]; s_ij += R[i][j - i + W]):
        // the 8 waves split the contraction over d (k-steps of 4), partial tiles meet in LDS.  A = Q[i = lane & 15][d], B = K[j = lane & 15][d]
        // (rows of pitch dk + 4: (dk + 4) mod 64 is a multiple of 4 and odd / 4 -> the 16 rows of a lane group and the 4 d of a k-step
        // hit 64 different banks).
        const bool near = W > 0 && j0 - i0 <= 15 + W && i0 - j0 <= 15 + W;  // (block-uniform) the block touches the band
        {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, racc = {0.f, 0.f, 0.f, 0.f};
          const int ro = (lane & 15) * dkp + (lane >> 4);
          const int nks = dk >> 2;
          for (int ks = wave; ks < nks; ks += PS_WAVES) {
            const float qa = Qs[ro + 4 * ks], kb = Ks[ro + 4 * ks];
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, kb, sacc, 0, 0, 0);
            if (near) racc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, Ek[ro + 4 * ks], racc, 0, 0, 0);  // (rows >= 2W + 1: never read)
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { Sp[(wave * 4 + r) * 64 + lane] = sacc[r]; Rp[(wave * 4 + r) * 64 + lane] = racc[r]; }
        }
        __syncthreads();
        PS_STAMP(4);
        float m_i = 0.f, l_i = 0.f;
        if (wave < 4) {  // (tid < 256) thread = (query i = tid >> 4, key j = tid & 15): the 16 lanes of a DPP row = the keys of query i
          const int i = tid >> 4, j = tid & 15;
          float sc = 0.f;
#pragma unroll
          for (int w = 0; w < PS_WAVES; ++w) sc += Sp[(w * 4 + (i & 3)) * 64 + (i >> 2) * 16 + j];
          const int rel = (j0 + j) - (i0 + i) + W;  // relative position index
          if (near) {
            const int rc = rel < 0 ? 0 : (rel > 15 ? 15 : rel);
            float rs = 0.f;
#pragma unroll
            for (int w = 0; w < PS_WAVES; ++w) rs += Rp[(w * 4 + (i & 3)) * 64 + (i >> 2) * 16 + rc];
            sc += (rel >= 0 && rel <= 2 * W) ? rs : 0.f;
          }
          const bool kok = j0 + j < L;
          sc = kok ? sc : -3.0e38f;
          m_i = ps_row_max(sc);
          const float p = kok ? __expf(sc - m_i) : 0.f;
          l_i = ps_row_sum(p);
          Ps[i * 17 + j] = p;
        }
        __syncthreads();
        PS_STAMP(5);
        // ---- O = P V (+ the relative-value band P_rel E_v, P_rel[i][r] = p_{i, j = i + r - W + i0 - j0}): wave w owns the 16 head
        // dimensions d0 = 16 w; A = P[i = lane & 15][j], B = V[j][d0 + (lane & 15)]
        if (wave * 16 < dk) {
          const int d0 = wave * 16, n = lane & 15, kk = lane >> 4;
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(Ps[n * 17 + 4 * ks + kk], Vs[(4 * ks + kk) * dkp + d0 + n], o, 0, 0, 0);
          if (near) {
            const int jb = n + (i0 - j0) - W + kk;  // key (tile-local) of relative position r = kk for query i = n; + 4 per k-step
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
              const int j = jb + 4 * ks, r = 4 * ks + kk;
              const float pr = Ps[n * 17 + (j < 0 ? 0 : (j > 15 ? 15 : j))];
              o = __builtin_amdgcn_mfma_f32_16x16x4f32((j >= 0 && j < 16 && r <= 2 * W) ? pr : 0.f, Ev[r * dkp + d0 + n], o, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // accumulator row 4 (lane >> 4) + r, column lane & 15
            const int i = 4 * kk + r;
            if (i0 + i < L) ll_store_off(ap, (unsigned)((((kt * Tp + i0 + i) * nh + hd) * dk2) + d0 + n) * 8u, o[r], epoch);
          }
        }
        if (wave < 4 && (tid & 15) == 0) {
          const int i = tid >> 4;
          if (i0 + i < L) {
            PS_G ll_t* dst = ap + (((long long)kt * Tp + i0 + i) * nh + hd) * dk2 + dk;
            ll_store(dst, m_i, epoch);
            ll_store(dst + 1, l_i, epoch);
          }
        }
      }
      PS_STAMP(3);
      continue;
    }

    continue;  // (no other kind exists)
    }
    // ==================================================================== matrix step
    // Written for INSTRUCTION COUNT: with two waves per SIMD that run the same code, a wave issues one instruction per ~8 cycles, and
    // the round-3 form of this step was ~1100 instructions per wave -- its 10 k cycles were issue time, not memory time (the round-4
    // trace: 1.06 k cycles between the staging barrier and the first MFMA without a single memory access in between).  Everything that
    // does not depend on the lane is kept on the scalar unit: the window column a thread gathers depends only on its WAVE (waves 0-3
    // take the even columns, 4-7 the odd ones), so the validity of a column is a scalar compare and a scalar branch, and the column's
    // address is a scalar base plus ONE lane offset shared by all ten loads.
    ps_prefetch(rv, tid, wave, lane, pre);
    __syncthreads();  // the previous step's readers of the LDS buffers are done
    {
      const int Cin = PR_I(rv, 1) & 0xffff, K = (PR_I(rv, 1) >> 16) & 0xf, ROW = 16 + K - 1;
      const int cps = PR_I(rv, 2), t0w = PR_I(rv, 3);
      const int n_u = PR_B(rv, 0) & 0x7f, nblk = (PR_B(rv, 0) >> 7) & 0xf, wstride = PR_B(rv, 1), ypitch = PR_B(rv, 2), rows_left = PR_B(rv, 3);
      const int pT = PR_B(rv, 4), n0 = PR_B(rv, 6) & 0xffff, gate_H = PR_B(rv, 7);
      // LDS byte offsets of this wave's tap units (one scalar load of eight dwords; K is 1, 3 or 5: persist_slice)
      int uoff[PS_MAXU];
#pragma unroll
      for (int i = 0; i < PS_MAXU; ++i) uoff[i] = ps_unit_tab.off[K >> 1][wave][i];
      const int cnt = n_u > wave ? (n_u - wave + PS_WAVES - 1) >> 3 : 0;  // tap units of this wave
      const unsigned umask = (1u << cnt) - 1u;                             // bit i: unit i exists (one scalar bit test per unit)
      float rsd0 = 0.f;  // residual cell of row block 0 (polled with the window)
      // ---- operand window [Cin][ROW] -> LDS (transposed: cells are column-major); thread = (channel c, window columns jh + 2 k)
      {
        const int jh = wave >> 2;                // (wave-uniform)
        const int c = (wave & 3) * 64 + lane;    // == tid & 255
        const bool cok = c < Cin;
        const int lim = (kf & PF_INMASK) ? L : Tp;
        constexpr int NG = PS_MAXROW / 2;
        const int nk = (ROW - jh + 1) >> 1;      // columns of this wave: jh + 2 k < ROW
        const int tj = t0w + jh;
        // which of this wave's columns exist: bit k set <=> k < nk and 0 <= tj + 2 k < lim  (one scalar bit test per column below)
        const int klo = tj < 0 ? (1 - tj) >> 1 : 0, khi_t = lim > tj ? (lim - tj + 1) >> 1 : 0, khi = khi_t < nk ? khi_t : nk;
        const unsigned kmask = khi > klo ? ((1u << khi) - 1u) & ~((1u << klo) - 1u) : 0u;
        float v[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) v[k] = 0.f;
        if (kf & PF_PLAIN_IN) {
          // plain floats [channels][pT] written by an earlier kernel; cps = +-1 (channel direction)
          const PS_G float* bp = PR_P(const float, rv, 0) + (long long)(cok ? c : 0) * cps * pT;
          const int limp = lim < pT ? lim : pT;
#pragma unroll
          for (int k = 0; k < NG; ++k) {
            const int t = tj + 2 * k;
            if (k < nk && t >= 0 && t < limp) v[k] = bp[t];  // (uniform condition; limp, not lim: kmask does not apply)
          }
#pragma unroll
          for (int k = 0; k < NG; ++k) v[k] = cok ? v[k] : 0.f;
        } else {
          const PS_G ll_t* bin = PR_P(const ll_t, rv, 0);
          const int acp = cps < 0 ? -cps : cps;
          const int cc = cok ? (cps < 0 ? Cin - 1 - c : c) : 0;  // (negative direction: the record's base points at the slice's LOWEST channel)
          unsigned lo = (unsigned)cc * 8u;
          const unsigned colb = (unsigned)acp * 8u;                 // bytes per window column
          const unsigned col0 = (unsigned)tj * colb;                // (wraps for tj < 0: those columns are never requested)
          const PS_G ll_t* res = PR_P(const ll_t, rv, 3);
          const bool rok = res != nullptr && wave < 4 && (tid & 15) < rows_left;
          unsigned ro = (unsigned)((tid >> 4) * PR_B(rv, 5) + (tid & 15)) * 8u;
          // cells that are not requested read as {0.0f, this epoch}: the check pass below needs no conditions (and every load of a
          // round is issued before the first check -- a check inside the load's own branch made hipcc wait for each load in turn)
          const ll_t zero_cell = (ll_t)epoch << 32;
          ll_t q[NG], qr = zero_cell;
#pragma unroll
          for (int k = 0; k < NG; ++k) q[k] = zero_cell;
          auto issue = [&]() {
            asm volatile("" : "+v"(lo), "+v"(ro));  // (addresses stay inside the loop body: no hoisted 64-bit pairs)
            if (cok) {
#pragma unroll
              for (int k = 0; k < NG; ++k)  // wave-uniform condition: a scalar bit test and branch; 32-bit offsets (a cell array is < 4 GiB)
                if (kmask & (1u << k)) q[k] = ll_load_off(bin, lo + (col0 + 2u * k * colb));
            }
            if (rok) qr = ll_load_off(res, ro);
          };
          auto check = [&]() -> bool {
            unsigned bad = ll_bad(qr, epoch);
#pragma unroll
            for (int k = 0; k < NG; ++k) bad |= ll_bad(q[k], epoch);
            return PS_PENDING(bad);
          };
          // (the first poll round BEFORE this step's own operand requests -- weights and epilogue vectors flying under its round trip --
          //  was measured again in round 4: c2 0.917 against 0.890 ms; the store -> visible latency of the producers is longer than the
          //  bookkeeping in front of the poll, so an earlier poll only adds a failed round)
          bool pending;
          do {
            issue();
            pending = check();
          } while (ps_again(cx, pending));
#pragma unroll
          for (int k = 0; k < NG; ++k) v[k] = ll_val(q[k]);
          rsd0 = ll_val(qr);
        }
        PS_STAMP(1); PS_REC_READY();
        // The poll has completed: this CU's memory queue is empty and ~4 k cycles of LDS and matrix work follow.  The moment to pull
        // the weights of this worker's NEXT matrix item towards it (into the XCD's L2; the workers of the other column tiles that
        // stream the same fragments sit on the same XCD, persist_plan.hip.h).  Measured neutral to slightly negative at c2 (round 4,
        // profiles/r4_persist_ab.txt): off by default, kept as a switch (VITS_PS_TUNE bit 0).
        if (call.tune & PS_TUNE_TOUCH) {
          const int nx = PR_B(rv, 0) >> 11;
          if (nx & 0x7f)
            ps_touch_weights(PR_P(const char, rv, 9), nx & 0x7f, (nx >> 7) & 0xf, (unsigned)((nx >> 11) & 0x1ff) << 10, wave, lane,
                             __builtin_amdgcn_readfirstlane((unsigned)(size_t)dma_sink));
        }
        if (cok) {
          float* tp = tile + c * PS_TP + jh;
#pragma unroll
          for (int k = 0; k < NG; ++k)
            if (k < nk) tp[2 * k] = v[k];
        }
      }
      __syncthreads();
      PS_STAMP(2);

      // ---- MFMA tiles of this item's 16-row blocks + epilogues
      const char* blb = reinterpret_cast<const char*>(tile + (lane >> 4) * PS_TP + (lane & 15));
      const PS_G float* wbase = PR_P(const float, rv, 6);
      for (int mi = 0; mi < nblk; ++mi) {
        if (mi > 0) __syncthreads();  // mred of the previous block has been read
        const float eb0 = pre.eb0, eb1 = pre.eb1, ec0 = pre.ec0, ec1 = pre.ec1;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (mi == nblk - 1) PS_STAMP(7);
        do {  // this wave's tap units, leaving at the first one it does not have (one scalar compare per unit)
#define PS_UNIT(i)                                                                                              \
          if (!(umask & (1u << (i)))) break;                                                                    \
          {                                                                                                     \
            const float* bp = reinterpret_cast<const float*>(blb + uoff[i]);                                    \
            const float b0 = bp[0], b1 = bp[4 * PS_TP], b2 = bp[8 * PS_TP], b3 = bp[12 * PS_TP];                \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pre.a[i][0], b0, acc0, 0, 0, 0);                        \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pre.a[i][1], b1, acc1, 0, 0, 0);                        \
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pre.a[i][2], b2, acc0, 0, 0, 0);                        \
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pre.a[i][3], b3, acc1, 0, 0, 0);                        \
          }
          PS_UNIT(0) PS_UNIT(1) PS_UNIT(2) PS_UNIT(3) PS_UNIT(4) PS_UNIT(5) PS_UNIT(6) PS_UNIT(7)
#undef PS_UNIT
        } while (0);
        if (mi == nblk - 1) PS_STAMP(4);
        // residual cells of this block: block 0's came with the window; later blocks (data of an older step: normally one round trip)
        float rsd = rsd0;
        const PS_G ll_t* res = PR_P(const ll_t, rv, 3);
        if (mi > 0 && res) {  // (every wave runs the loop -- a poll loop under a per-lane condition makes the poll state per-lane values)
          unsigned ro = (unsigned)((tid >> 4) * PR_B(rv, 5) + mi * 16 + (tid & 15)) * 8u;
          const bool rok = wave < 4 && mi * 16 + (tid & 15) < rows_left;
          bool pending;
          do {
            asm volatile("" : "+v"(ro));
            ll_t q = 0;
            if (rok) q = ll_load_off(res, ro);
            rsd = ll_val(q);
            pending = PS_PENDING(rok ? ll_bad(q, epoch) : 0u);
          } while (ps_again(cx, pending));
        }
        // the weight registers are free: the NEXT row block's fragments and epilogue vectors fly under this block's reduction and
        // epilogue (requested at the top of the next iteration -- behind the epilogue and a barrier -- their whole latency was exposed)
        if (mi + 1 < nblk) {
          int i0, i1;
          ps_epi_idx(kf, rows_left, gate_H, tid, mi + 1, i0, i1);
          const PS_G float* b = PR_P(const float, rv, 7);
          const PS_G float* c = PR_P(const float, rv, 8);
          pre.eb0 = b[i0]; pre.eb1 = b[i1]; pre.ec0 = c[i0]; pre.ec1 = c[i1];
          ps_load_weights(wbase + (size_t)(mi + 1) * wstride, n_u, wave, lane, pre.a);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mred[(wave * 4 + r) * 64 + lane] = acc0[r] + acc1[r];
        __syncthreads();
        if (mi == nblk - 1) PS_STAMP(6);
        if (kf & PF_GATE) {
          // packed 16-row block = [8 tanh rows | 8 sigmoid rows] of channels 8 * mb .. 8 * mb + 7 (commons.py:100-107)
          if (wave < 2) {  // (tid < 128)
            const int ch = tid & 7, col = tid >> 3;
            float at = 0.f, as = 0.f;
#pragma unroll
            for (int w = 0; w < PS_WAVES; ++w) {
              at += mred[(w * 4 + (ch & 3)) * 64 + (ch >> 2) * 16 + col];
              as += mred[(w * 4 + (ch & 3)) * 64 + ((ch >> 2) + 2) * 16 + col];
            }
            const float tv = tanhf(at + eb0 + ec0);
            const float sv = 1.0f / (1.0f + __expf(-(as + eb1 + ec1)));
            ll_store_off(PR_P(ll_t, rv, 1), (unsigned)(col * ypitch + mi * 8 + ch) * 8u, tv * sv, epoch);  // (p1 points at channel 8 * first block)
          }
        } else if (wave < 4) {  // (tid < 256)
          const int row = tid & 15, col = tid >> 4;  // rows fastest: a column's 16 cells are one 128-byte segment
          float v = 0.f;
#pragma unroll
          for (int w = 0; w < PS_WAVES; ++w) v += mred[(w * 4 + (row & 3)) * 64 + (row >> 2) * 16 + col];
          const int rl = mi * 16 + row;
          const int t = n0 + col;
          if (rl < rows_left) {
            v += eb0 + ec0;
            if (kf & PF_RELU) v = v > 0.f ? v : 0.f;
            if ((kf & PF_OUTMASK) && t >= L) v = 0.f;
            v += rsd;
            if (kf & PF_SPLINE) hb[rl * 16 + col] = v;
            else {
              PS_G ll_t* yo = PR_P(ll_t, rv, 1);
              PS_G float* yp = PR_P(float, rv, 2);
              if (yo) ll_store_off(yo, (unsigned)(col * ypitch + rl) * 8u, v, epoch);
              if (yp && t < pT) *(PS_G float*)((PS_G char*)yp + (unsigned)(rl * pT + col) * 4u) = v;
            }
          }
        }
      }
      if ((kf & PF_ZINIT) && wave == 0 && lane < 32) {
        // z = randn * noise_scale_w (models.py:96): injected noise or the Philox stream of dp_init_z_kernel
        const int c = tid >> 4, t = n0 + (tid & 15);
        float nsw = call.nsw;
        unsigned long long seed = call.seed;
        if (call.dv) { nsw = call.dv->scales[2]; seed = call.dv->seed; }
        float e = 0.f;
        if (t < T) e = call.noise ? call.noise[(long long)c * T + t]
                                  : (call.solo ? philox_normal(call.item_seeds ? call.item_seeds[0] : seed, 1, (uint32_t)c, (uint32_t)t)
                                               : philox_normal(seed, 1, (uint32_t)c, (uint32_t)t));
        ll_store(PR_P(ll_t, rv, 5) + (long long)c * Tp + t, e * nsw, epoch);
      }
      if (kf & PF_SPLINE) {
        // Inverse rational-quadratic spline of the tile's 16 columns (transforms.py:55-177), the arithmetic of spline_inverse_elem
        // (kernels_misc.hip.h) spread over the workgroup: one thread per column ran ~23 k cycles (20 expf and 20 divisions in a
        // dependent chain); here the exponentials are one per thread, the bin softmax / cumulative sums are DPP row operations, and 16
        // threads finish (bin search, quadratic).
        __syncthreads();  // h complete
        // z of the tile's columns (cells of a much older step): requested now, checked where they are used -- one round trip under
        // the exponentials and scans instead of behind them
        ll_t zq0 = 0, zq1 = 0;
        if (wave == 0) {
          const int t = n0 + (lane & 15);
          const int x0r = (PR_B(rv, 6) >> 16) & 1;
          const PS_G ll_t* zin = PR_P(const ll_t, rv, 4);
          if (lane < 16 && t < L) {
            zq0 = ll_load_off(zin, (unsigned)(x0r * Tp + t) * 8u);
            zq1 = ll_load_off(zin, (unsigned)((1 - x0r) * Tp + t) * 8u);
          }
        }
        const int nb = prog->nb;
        const float bound = prog->bound, isd = prog->inv_sqrt_d;
        float* sc = mred;  // [2][17][16] cumulative widths / heights (knots)
        {
          // softmax over the bins, cumulative sum and knots (transforms.py:96-131) with the 16 lanes of a DPP row = the bins of ONE
          // column: thread = (which = wave >> 2: widths / heights, column = 4 (wave & 3) + (lane >> 4), bin k = lane & 15).  Row
          // maximum, row sum and the inclusive prefix are DPP row operations (no LDS, no loop): the serial form this replaces (32
          // threads walking the bins: two dependent LDS reads per bin, then a 16-thread bin search) took 11 k cycles per spline step.
          const int which = wave >> 2, col = 4 * (wave & 3) + (lane >> 4), k = lane & 15;
          const bool kok = k < nb;
          const float u = kok ? hb[(which * nb + k) * 16 + col] * isd : -3.0e38f;
          const float mx = ps_row_max(u);
          const float e = kok ? expf(u - mx) : 0.f;
          const float sum = ps_row_sum(e);
          const float mn = 1e-3f;  // min_bin_width == min_bin_height
          float c = kok ? mn + (1.f - mn * nb) * (e / sum) : 0.f;
          c += ps_row_shr<1>(c);
          c += ps_row_shr<2>(c);
          c += ps_row_shr<4>(c);
          c += ps_row_shr<8>(c);
          float* cdst = sc + which * 17 * 16 + col;
          if (k == 0) cdst[0] = -bound;
          if (kok) cdst[(k + 1) * 16] = k == nb - 1 ? bound : 2.f * bound * c - bound;
        }
        __syncthreads();
        if (wave == 0) {  // one column per lane (lanes >= 16 ride along in the wave-uniform poll)
          const int col = lane & 15, t = n0 + col;
          const int x0r = (PR_B(rv, 6) >> 16) & 1, x1r = 1 - x0r, ea_row = (PR_B(rv, 6) >> 20) & 1;
          const PS_G ll_t* zin = PR_P(const ll_t, rv, 4);
          PS_G ll_t* zout = PR_P(ll_t, rv, 5);
          const bool need = lane < 16 && t < L;
          unsigned o0 = (unsigned)(x0r * Tp + t) * 8u, o1 = (unsigned)(x1r * Tp + t) * 8u;
          float z0 = ll_val(zq0), z1 = ll_val(zq1);
          bool pending = PS_PENDING(need ? (ll_bad(zq0, epoch) | ll_bad(zq1, epoch)) : 0u);
          while (ps_again(cx, pending)) {  // (not yet there: poll)
            asm volatile("" : "+v"(o0), "+v"(o1));
            ll_t q0 = 0, q1 = 0;
            if (need) { q0 = ll_load_off(zin, o0); q1 = ll_load_off(zin, o1); }
            const unsigned bad = need ? (ll_bad(q0, epoch) | ll_bad(q1, epoch)) : 0u;
            z0 = ll_val(q0); z1 = ll_val(q1);
            pending = PS_PENDING(bad);
          }
          if (lane < 16) {
            float v0 = 0.f, v1 = 0.f;
            if (t < L) {
              v0 = z0;
              v1 = z1;
              const float y = z1;
              if (y >= -bound && y <= bound) {  // identity outside the interval (transforms.py:65-77)
                const float* cw = sc + col;
                const float* ch = sc + 17 * 16 + col;
                int bin = -1;
                for (int k = 0; k <= nb; ++k) {
                  const float loc = ch[k * 16] + (k == nb ? 1e-6f : 0.f);
                  if (y >= loc) bin++;
                }
                bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
                const float in_cw = cw[bin * 16], in_w = cw[(bin + 1) * 16] - in_cw;
                const float in_ch = ch[bin * 16], in_h = ch[(bin + 1) * 16] - in_ch;
                const float min_d = 1e-3f;
                const float cst = spline_cst;
                const float ud0 = (bin == 0) ? cst : hb[(2 * nb + bin - 1) * 16 + col];
                const float ud1 = (bin == nb - 1) ? cst : hb[(2 * nb + bin) * 16 + col];
                const float d0 = min_d + softplus_f(ud0), d1 = min_d + softplus_f(ud1);
                const float delta = in_h / in_w;
                const float t1 = (y - in_ch) * (d0 + d1 - 2.f * delta);
                const float qa = t1 + in_h * (delta - d0);
                const float qb = in_h * d0 - t1;
                const float qc = -delta * (y - in_ch);
                const float disc = qb * qb - 4.f * qa * qc;
                const float root = (2.f * qc) / (-qb - sqrtf(disc));
                v1 = root * in_w + in_cw;
              }
            }
            if (zout) {
              ll_store(zout + (long long)x0r * Tp + t, v0, epoch);
              ll_store(zout + (long long)x1r * Tp + t, v1, epoch);
            }
            if ((kf & PF_LAST) && t < T) {
              const float zz = ea_row == x0r ? v0 : v1;
              const float lw = t < L ? (zz - ea_m) * ea_is : 0.f;
              ((PS_G float*)prog->logw)[t] = lw;
              if (prog->logw_cells) ll_store((PS_G ll_t*)prog->logw_cells + t, lw, epoch);
            }
          }
        }
      }
    }
    PS_STAMP(3);
  }
  // ---- the last worker to finish publishes the epoch (every worker read it before doing anything else)
  __syncthreads();
  if (tid0 == 0) {
    const bool timed_out = cx.aborted || __hip_atomic_load(&call.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (timed_out) atomicOr((int*)prog->err, PS_ERR_TIMEOUT);
    const unsigned old = atomicAdd(&call.ctl->done, 1u);
    if (old == gridDim.x - 1) {
      if (!timed_out) atomicAdd(call.dbg + 1, 1);  // (diagnostics: tests assert that a call really took the persistent path)
      __hip_atomic_store(&call.ctl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&call.ctl->abort, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&call.ctl->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- packed per-thread parameters of a step (built once per model: persist_plan.hip.h): dst[row][k] = src[k][row + add[k]] or 0
struct PsPackArgs { const float* src[8]; int add[8]; int len[8]; };
__global__ void ps_pack_kernel(float* dst, PsPackArgs a, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 8) return;
  const int row = i >> 3, k = i & 7;
  const int idx = row + a.add[k];
  dst[i] = (a.src[k] && idx >= 0 && idx < a.len[k]) ? a.src[k][idx] : 0.f;
}
