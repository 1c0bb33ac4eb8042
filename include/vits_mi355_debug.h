/* vits_mi355_debug.h -- TEST HOOKS of libvits_mi355.so.  NOT part of the installed ABI: a deployment ships include/vits_mi355.h and
 * include/stts_mi355.h only; this header exists for tests/, tools/ and bench.py.
 *
 * Scope of a hook (round 6):
 *   - kernel-selection / path switches (force_tile, attention_impl, fast_path, ks_waves, ln_stats, wn_fold, conv_wp, conv_sp,
 *     no_bf16x3, tail_impl, poison_workspace) are THREAD-LOCAL: they change what the CALLING thread's engine calls launch and nothing
 *     another thread is running (the engine runs every call on the caller's thread);
 *   - the persistent-program switches (persist, persist_spin, persist_rearm_ms) are process-wide on purpose: the programs are a
 *     per-device resource with one owner at a time; they are atomics and take effect for calls that start afterwards.
 */
#ifndef VITS_MI355_DEBUG_H
#define VITS_MI355_DEBUG_H
#include "vits_mi355.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: 0 = choose the conv kernel by problem size (default), 1 = always the big-tile kernel,
 * 2 = always the K-split small-N kernel, 3 = the small-tile (16x16x4 MFMA, LDS-staged) kernel wherever a launch is
 * eligible for it.  Process-wide. */
void vits_debug_force_tile(int mode);
/* Test hook: 0 = by sequence length (default: 16-query MFMA kernel up to T = 512, 32-query MFMA flash kernel beyond),
 * 1 = the scalar-VALU attention kernel, 2 = always the 32-query kernel, 3 = always the 16-query kernel. */
void vits_debug_attention_impl(int impl);
/* Test hook: 1 (default) = vits_synthesize replays captured hipGraphs over bucketed shapes when no noise tensor is
 * injected; 0 = always the eager path (one launch per kernel, exact-size workspace).  Both give the same samples. */
void vits_debug_fast_path(int on);
/* Test hook: waves per workgroup of the K-split conv kernel: 0 = size heuristic (default), 4 / 8 / 16 forced. */
void vits_debug_ks_waves(int nw);
/* Test hook: 1 (default) = the folded encoder LayerNorms take their channel statistics from the producing conv's epilogue,
 * 0 = every consumer workgroup recomputes them. */
void vits_debug_ln_stats(int on);
/* Bit mask of the single-utterance stages (B = 1, T <= 256) that run as ONE persistent kernel each, with in-band ("LL cell")
 * exchange between their steps (csrc/persist.hip.h): 1 duration predictor, 2 text encoder, 4 flow.  Default 7; 0: the launch-per-layer
 * path everywhere (the A/B reference, and the batch path).  Setting the mask also ends a timeout's off interval at once. */
void vits_debug_persist(int mask);
/* Test hook: poll rounds after which a worker of a persistent program gives up (0 = the default bound, 2^18).  A timeout
 * switches the persistent programs off for a bounded interval (below) and the host entry points run the call again on the launch
 * path (the caller sees a slower call, not an error); an asynchronous device session reports VITS_ERR_DEVICE once. */
void vits_debug_persist_spin(int limit);
/* WHEN a host call takes the persistent programs (process-wide; environment VITS_PERSIST_WHEN): a single-utterance call runs them when at
 * most `max_others_in_flight` OTHER host calls are in flight on its device as it starts (and the device's program token is free).  Default 1
 * (round 6): next to several other calls' kernels a program is placed late and its spinning workers slow everybody down -- 4 clients
 * 1910 -> 2030 requests/s, p50 2.4 -> 1.9 ms on launches only -- while with 2 clients "one on the programs, one on launches" wins (1460
 * against 1240 requests/s); profiles/r6_owners.txt.  0 = only a call that starts alone; -1 = whenever the token is free (rounds 3-5).
 * Device sessions (vits_session_*) take the token for their lifetime either way. */
void vits_debug_persist_when(int max_others_in_flight);
/* Test hook: base re-arm interval in milliseconds (0 = VITS_PERSIST_REARM_MS or 1000). */
void vits_debug_persist_rearm_ms(int ms);
/* Test hook: persistent-program launches of this model that ran to completion (no timeout) since vits_create; -1 on error.
 * (The bound of vits_debug_persist_spin is a device word the kernel reads at run time: captured graphs follow it.) */
int vits_debug_persist_runs(vits_model* m);
/* Test hook: 1 (default) = WaveNet tail of the coupling layers in folded form (gate outputs of all layers kept, one conv =
 * post o sum of skip halves), 0 = per-layer res/skip accumulation + post as the reference executes it.  Same results to rounding. */
void vits_debug_wn_fold(int on);
/* Test hook: wave-pipelined decoder conv kernel (conv_wp_kernel): 0 = by size (default), 1 = never, 2 = whenever eligible. */
void vits_debug_conv_wp(int mode);
/* Test hook, host arithmetic only (no device is touched): the per-layer limits of the decoder in a ragged batch whose items continue into
 * the padding like the reference's padded batch (engine.hip decoder_needs) -- how many columns beyond an item's end each launch still
 * produces.  out[0] = frames of z the decoder reads beyond an item's end, [1] conv_pre's output limit, [2] conv_post's, [3] columns the
 * iSTFT / PQMF tail reads, then per upsampling stage: the polyphase launch's limit (input positions), c1 limits [n_resd], c2 limits
 * [n_resd].  Returns the number of values (at most `cap` are written), or a negative error. */
int vits_debug_decoder_needs(const vits_hparams* hp, int32_t* out, int32_t cap);
/* Test hook: software-pipelined 64 x 64 conv kernel (conv_sp_kernel, csrc/conv_sp.hip.h): -1 = environment / default (by grid size),
 * 0 = never, 1 = by grid size, 2 = wherever a launch is eligible for it. */
void vits_debug_conv_sp(int mode);
/* Test hook: stream-K schedule of the 64 x 64 pipelined tile (conv_sk_kernel, csrc/conv_sk.hip.h; a measured prototype, 8-17 % SLOWER than
 * the plain launch on s8 / s16, profiles/r6_sk_ab.txt): -1 = environment VITS_SK (default 0 = off), 0 = off, 1 = the launches conv_sp_kernel<STORE>
 * takes by size, 2 = wherever it is eligible.  A session gets the kernel's exchange buffers the next time its workspace is laid out. */
void vits_debug_conv_sk(int mode);
/* Test hook: 1 = a conv_precision == 1 model runs its fp32 kernels instead of the split-bf16 variant (same weights, A/B). */
void vits_debug_no_bf16x3(int on);
/* Test hook: 0 = fused exp/sin + iSTFT + PQMF tail kernel (default), 1 = the separate istft / pqmf kernels. */
void vits_debug_tail_impl(int impl);
/* Test hook: fill every newly laid-out workspace with NaN bit patterns (stale-padding detector). */
void vits_debug_poison_workspace(int on);

/* Shader clock under load (round 6): launches `n` one-wave workgroups on a stream of the library's own that sit on the device for
 * `duration_us` and compare the shader-clock counter (s_memtime) with the constant 100 MHz wall clock (s_memrealtime); ghz[i] = the clock
 * workgroup i's CU ran at while whatever else is on the device (a bench loop on another stream) was running.  Blocks until the probe is
 * done; the first call per device creates the probe's stream and buffer (call it once BEFORE the work it is to watch: allocation waits
 * for work in flight).  Returns the number of values written or a negative error.
 * What it found (profiles/r6_bt_clock.txt, r6_clock_in_forward.txt): back-to-back dense conv launches on N(0,1) operands pull the clock
 * down to 1.86 - 2.11 GHz, the forwards of the bench (synthetic weights of trained-model scale) run at 2.35 - 2.40: the 2.4 GHz peak is
 * the right yardstick for the bench lines, and a microbenchmark on random data is not a proxy for them. */
int vits_debug_clock_probe(int device, int32_t duration_us, double* ghz, int32_t n);

#ifdef __cplusplus
}
#endif
#endif
