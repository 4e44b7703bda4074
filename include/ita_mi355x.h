/*
 * ita_mi355x.h -- C ABI of libita_mi355x.so, the MI355X (gfx950) replacement for the
 * reference's ITA custom-dispatch plugin.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes (no torch / HIP types in
 * the signatures; a stream is passed as `void*` holding a hipStream_t, NULL = default stream)
 * and -- except the two drop-in symbols whose reference prototype is `void` -- returns an
 * ita_status.  All file:line citations are relative to the reference repository.
 *
 * Boundary being replaced
 *   samples/inference_udp_FPGA_custom_dispatch/plugin/ITA_dispatch.c:17-27   ITASelfAttention_workgroup
 *   samples/inference_udp_FPGA_custom_dispatch/plugin/ITA_dispatch.c:31-57   ..._workgroup_expanded
 *   samples/inference_udp_FPGA_custom_dispatch/plugin/ITA_spec.mlir:6-9,21-33 2 bindings, 0 constants, 1 workgroup
 *   samples/inference_udp_FPGA_custom_dispatch/main.cpp:171-201               module.main_graph 5-in / 3-out
 *   tests/export_onnx_for_FPGA.py:71-80                                       I/O names of main_graph
 */
#ifndef ITA_MI355X_H_
#define ITA_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITA_MI355X_ABI_VERSION 1

typedef struct ita_context* ita_handle;

typedef enum ita_status {
  ITA_OK = 0,
  ITA_ERR_INVALID_ARG = -1,
  ITA_ERR_BAD_BLOB = -2,      /* magic / bounds / missing tensor */
  ITA_ERR_NO_WEIGHTS = -3,    /* compute call before ita_load_weights */
  ITA_ERR_UNSUPPORTED = -4,   /* dims the kernels are not built for */
  ITA_ERR_HIP = -5,           /* a HIP runtime call failed; see ita_error_string */
  ITA_ERR_NO_DEVICE = -6,
  ITA_ERR_NOT_BOUND = -7      /* drop-in symbol called without ita_bind_dispatch */
} ita_status;

typedef enum ita_image_dtype { ITA_IMAGE_F32 = 0, ITA_IMAGE_U8 = 1 } ita_image_dtype;
typedef enum ita_dispatch_dtype { ITA_DISPATCH_F16 = 0, ITA_DISPATCH_F32 = 1 } ita_dispatch_dtype;

/* ---- lifetime --------------------------------------------------------------------------- */
int ita_abi_version(void);
/* Creates a context on HIP device `device_ordinal` (-1: the calling thread's current device). */
int ita_create(ita_handle* out, int device_ordinal);
int ita_destroy(ita_handle h);
/* Uploads an "ITAW0001" blob (ita_weights.h) from host memory; replaces any earlier weights.
 * The reference pre-loads weights into the accelerator out of band
 * (docs/HOW-TO-run-the-full-project-workflow.md:55); this is that step. */
int ita_load_weights(ita_handle h, const void* blob, size_t nbytes);
/* The structural check ita_load_weights runs first (ita_weights.h: ita_blob_validate): header, table bounds, and the exact
 * dtype / byte size of every tensor the loader knows, derived from E / P / F / num_layers.  Host only, needs no GPU.
 * bad_name32 (char[32], may be NULL) receives the offending tensor's name. */
int ita_validate_blob(const void* blob, size_t nbytes, char* bad_name32);
/* Pre-sizes the internal workspace for batches up to max_batch so that the compute calls below allocate nothing, and
 * PINS it: after ita_reserve a larger batch is an error (ITA_ERR_INVALID_ARG) instead of a silent reallocation -- HIP
 * graphs captured over this handle and ita_vitlstm_front / _back pairs hold raw pointers into the workspace.  Call
 * ita_reserve again (it synchronises the device) to move to a larger size.  Without any ita_reserve the compute calls
 * grow the workspace on demand (an implicit device synchronisation; refused inside a stream capture).
 *
 * Concurrency: a handle owns ONE workspace and one set of profiling state -- at most one call may be in flight per
 * handle at a time, on one stream (calls on the same stream are ordered and safe).  Use one handle per stream or per
 * thread; the weights are small (about 25 MB with the derived buffers). */
int ita_reserve(ita_handle h, int max_batch);
int ita_get_dims(ita_handle h, int* E, int* S, int* P, int* F, int* H, int* num_layers);

/* ---- errors ------------------------------------------------------------------------------ */
int ita_last_error(void);                 /* status of the calling thread's last failing call */
const char* ita_error_string(void);       /* human readable detail for ita_last_error()        */

/* ---- hot path: device pointers, asynchronous on `stream` ---------------------------------- */

/* ITASelfAttention_QAT.forward (models/ITA/QAT/layers.py:101-127): x (B,128,E) f32 -> y (B,128,E)
 * f32 = dequantised out_proj.  layer < num_layers. */
int ita_mha_int8(ita_handle h, int layer, const float* x_dev, float* y_dev, int batch, void* stream);

/* As above, also writing the per-stage integer tensors (any pointer may be NULL):
 * x_q (B,128,E) s8, Q/K/V (B,128,P) s8, logits (B,128,128) s8, probs (B,128,128) u8,
 * ctx (B,128,P) s8, out_q (B,128,E) s8.  Used by the parity tests
 * (the reference hooks the same tensors: tests/export_and_validation_W_B.py:25-102). */
typedef struct ita_mha_taps {
  int8_t *x_q, *Q, *K, *V, *logits;
  uint8_t* probs;
  int8_t *ctx, *out_q;
} ita_mha_taps;
int ita_mha_int8_taps(ita_handle h, int layer, const float* x_dev, float* y_dev, int batch,
                      const ita_mha_taps* taps, void* stream);

/* The same block at the accelerator's own boundary: int8 codes in (what `quant` produces, layers.py:103), int8 out_proj
 * codes out (what `dequant` consumes, layers.py:125) -- x_q, out_q (B,128,E) s8 device pointers.  This is the form
 * SURVEY.md section 8(d) prices BASELINE config 2 on (32 KiB of HBM traffic per E = 128 frame instead of 128 KiB).
 * Equal, byte for byte, to the x_q -> out_q taps of ita_mha_int8_taps. */
int ita_mha_q8(ita_handle h, int layer, const int8_t* x_q_dev, int8_t* out_q_dev, int batch, void* stream);

/* The same block on LONG token sequences: x_q, out_q (B, seq_len, E) s8 with seq_len a multiple of 128 (E = 128 graphs only) --
 * BASELINE config 5 as it is worded (480 x 720 input, 64x patch-token blow-up: seq_len = 8192).  Same arithmetic as above
 * (layers.py:106-123, ITA_softmax.py:51-61 over a row of seq_len logits); the logits are never materialised: three sweeps over
 * the key tiles per query tile (row maximum, row sum, probabilities -> A.V), Q K^T recomputed in each.  Its Q / K / V^T
 * workspace (3 x 192 bytes per token) belongs to the handle and grows on demand (not inside a stream capture). */
int ita_mha_long_q8(ita_handle h, int layer, const int8_t* x_q_dev, int8_t* out_q_dev, int batch, int seq_len, void* stream);

/* ITAFeedForward_QAT.forward (models/ITA/QAT/layers.py:61-75). */
int ita_ffn_int8(ita_handle h, int layer, const float* x_dev, float* y_dev, int batch, void* stream);
typedef struct ita_ffn_taps { int8_t *x_q, *h, *out_q; } ita_ffn_taps;
int ita_ffn_int8_taps(ita_handle h, int layer, const float* x_dev, float* y_dev, int batch,
                      const ita_ffn_taps* taps, void* stream);

/* One encoder layer as the model wires it (QAT/model.py:100-113):
 * y = LN2(x1 + ffn(x1)),  x1 = LN1(x + mha(x)).  x_dev and y_dev may alias. */
int ita_encoder_layer(ita_handle h, int layer, const float* x_dev, float* y_dev, int batch, void* stream);

/* OverlapPatchMerging (models/ITA/QAT/layers.py:39-45): image (B,60,90) -> tokens (B,128,E).
 * ITA_IMAGE_F32: pixels already scaled to [0,1] (the reference host's float(pixel) / 255.0f, main.cpp:168-169).
 * ITA_IMAGE_U8: the 8-bit wire codes as they arrive (main.cpp:37,166); the division by 255 is folded, with the 1/256 of
 * the exact integer bilinear blend, into the conv weights -- the same linear map rounded once per weight, so the tokens
 * of the two forms agree to ~1e-6, not bit for bit (oracle/ita_oracle.c: ita_oracle_tokenizer_u8 / ita_oracle_tokenizer). */
int ita_tokenizer(ita_handle h, const void* image_dev, int image_dtype, float* tokens_dev, int batch, void* stream);

/* Fusion tail (QAT/model.py:116-121): x (B,128,E) -> (B,9,16,32) flattened, row stride 4608. */
int ita_fusion_tail(ita_handle h, const float* x_dev, float* feat_dev, int batch, void* stream);

/* module.main_graph with a leading batch (main.cpp:171-201; tests/export_onnx_for_FPGA.py:71-80):
 *   image (B,1,60,90) f32 or u8 wire frames, additional_data (B,1), quat_data (B,4),
 *   hidden_in_h / hidden_in_c (3,B,128)  ->  output (B,3), hidden_out_h / hidden_out_c (3,B,128).
 * hidden_out_* may alias hidden_in_* (in-place state).  optional float taps (may be NULL): tokens/x1/x2 (B,128,E),
 * feat (B,4608), dec (B,512). */
typedef struct ita_forward_taps { float *tokens, *x1, *x2, *feat, *dec; } ita_forward_taps;
int ita_vitlstm_forward(ita_handle h, const void* image_dev, int image_dtype, const float* additional_data_dev,
                        const float* quat_data_dev, const float* hidden_in_h_dev, const float* hidden_in_c_dev,
                        float* output_dev, float* hidden_out_h_dev, float* hidden_out_c_dev, int batch,
                        const ita_forward_taps* taps, void* stream);

/* The part of the graph behind the encoder on its own (QAT/model.py:116-130): x2 (B,128,64) f32 = the last LayerNorm2
 * output -> fusion tail -> decoder -> cat -> 3 LSTM layers -> fc, in the arithmetic ita_set_tail_mode selects. */
int ita_vitlstm_tail(ita_handle h, const float* x2_dev, const float* additional_data_dev, const float* quat_data_dev,
                     const float* hidden_in_h_dev, const float* hidden_in_c_dev, float* output_dev,
                     float* hidden_out_h_dev, float* hidden_out_c_dev, int batch, void* stream);

/* The same graph cut in two, for software pipelining ACROSS time steps.  A time step's image-only part
 * (tokenizer, encoder, folded tail GEMM) does not depend on the LSTM state, so step t+1's front can run on
 * one stream while step t's back (LSTM layers + fc, small latency-bound kernels) runs on another:
 *     front(t)  on stream F: image -> gate partials in internal buffer `buf` (0 .. ITA_PART_BUFFERS-1)
 *     back(t)   on stream B: partials[buf] + (h, c) -> velocity, new (h, c); must wait for front(t)
 * The caller orders them with events: back(t) after front(t); front(t') after back(t) when it reuses t's
 * buffer.  With buf = t % ITA_PART_BUFFERS the front stream does not wait for the back stream inside a window of
 * ITA_PART_BUFFERS steps.  Measured at 128 frames per step: front alone 32.6 us, back alone 17.3 us, one stream 50 us,
 * two streams 43 us (HIP graph of 8 steps, or ita_vitlstm_pipelined) -- the GPU overlaps about 7 of the 17 us.
 * front + back on one stream equals ita_vitlstm_forward.  Needs tail mode 1. */
#define ITA_PART_BUFFERS 8
int ita_vitlstm_front(ita_handle h, const void* image_dev, int image_dtype, int batch, int buf, void* stream);
/* the same, and records `encoder_done_event` (a hipEvent_t, may be NULL) on `stream` between the encoder and the
 * folded GEMM.  The encoder kernel owns every CU (one persistent workgroup each, all of LDS and VGPRs): small
 * kernels launched next to it only delay some of its workgroups.  Making back(t) wait for this event of front(t+1)
 * puts the LSTM kernels of step t next to the GEMM of step t+1, which leaves room for them. */
int ita_vitlstm_front_ev(ita_handle h, const void* image_dev, int image_dtype, int batch, int buf, void* stream,
                         void* encoder_done_event);
/* The front cut once more, for a THREE-stage pipeline: encoder(t+2) | folded GEMM(t+1) | LSTM + fc(t) on three streams.
 * ita_vitlstm_encode writes the encoder output planes into plane set `plane_set` (0 or 1), ita_vitlstm_fold multiplies
 * that set into partial buffer `buf`; encode + fold on one stream with plane_set 0 equals ita_vitlstm_front.  The caller
 * orders them with events: fold(t) after encode(t); encode(t') after fold(t) when it reuses t's plane set; back(t) after
 * fold(t); fold(t') after back(t) when it reuses t's partial buffer.  Measured from one HIP graph of 8 steps
 * (host.PipelinedSteps): 128 frames per step 38.2 us against 43.0 us for the two-stage form, 64 frames 36.1 / 39.7, one frame
 * 31.7 / 34.0. */
int ita_vitlstm_encode(ita_handle h, const void* image_dev, int image_dtype, int batch, int plane_set, void* stream);
int ita_vitlstm_fold(ita_handle h, int batch, int plane_set, int buf, void* stream);
int ita_vitlstm_back(ita_handle h, const float* additional_data_dev, const float* quat_data_dev,
                     const float* hidden_in_h_dev, const float* hidden_in_c_dev, float* output_dev,
                     float* hidden_out_h_dev, float* hidden_out_c_dev, int batch, int buf, void* stream);
/* n_steps consecutive time steps of the same `batch` streams, pipelined by this library on the caller's two (distinct,
 * non-default) streams: front(t+1) on stream_front overlaps back(t) on stream_back, ordered with events; the state
 * ping-pongs between state_h/state_c (3,batch,128; in: the incoming state, out: the state after the last step) and an
 * internal copy, so no step updates it in place.  image[t] / additional_data[t] / quat_data[t] / output[t] are HOST arrays
 * of n_steps device pointers.  Joins on stream_front: work enqueued there afterwards sees every output.  The calling
 * thread enqueues about six launches per step; it returns without waiting for the GPU. */
int ita_vitlstm_pipelined(ita_handle h, const void* const* image_dev, int image_dtype, const float* const* additional_data_dev,
                          const float* const* quat_data_dev, float* state_h_dev, float* state_c_dev, float* const* output_dev,
                          int batch, int n_steps, void* stream_front, void* stream_back);

/* Serving form of the same graph: the LSTM state of `num_slots` independent streams lives in two
 * persistent device arrays state_h / state_c of shape (3, num_slots, 128); frame b of the batch belongs to
 * stream slot_idx[b] (device int array, all distinct within one call) and updates that stream's state
 * in place.  This is how the reference host carries (h, c) from frame to frame (main.cpp:143-148,217-221),
 * generalised from one stream to many.  Needs tail mode 1. */
int ita_vitlstm_forward_slots(ita_handle h, const void* image_dev, int image_dtype, const float* additional_data_dev,
                              const float* quat_data_dev, float* state_h_dev, float* state_c_dev,
                              const int* slot_idx_dev, int num_slots, float* output_dev, int batch, void* stream);

/* Arithmetic of the float tail (fusion conv, decoder, LSTM) inside ita_vitlstm_forward:
 *   1 (default)  fusion conv, decoder and LSTM layer 0's input projection folded into one matrix at load
 *                time; all tail GEMMs on f16 MFMA with split-precision (hi+lo) operands: within 1e-5 of the
 *                f32 graph (task tolerance 1e-4).  The decoder output is never materialised: taps->dec is
 *                left untouched in this mode.
 *   0            the f32 kernels in the CPU oracle's operation order: equal to the oracle bit for bit.
 * hidden_out_* may alias hidden_in_* in both modes. */
int ita_set_tail_mode(ita_handle h, int mode);

/* ---- per-stage timing (bench.py's roofline leg) -------------------------------------------- */
/* Between ita_profile_begin and ita_profile_end every ita_vitlstm_forward call records HIP
 * events on ITS OWN stream around each stage; ita_profile_end synchronises them and returns
 * the summed device time per stage in milliseconds and the number of forwards covered.
 * Stages: 0 tokenizer, 1 int8 MHA (+LN1), 2 int8 FFN (+LN2), 3 fusion tail, 4 decoder GEMM,
 * 5 LSTM + fc.  At most max_forwards calls are recorded (later ones run unprofiled). */
#define ITA_NUM_STAGES 6
int ita_profile_begin(ita_handle h, int max_forwards);
/* Sampled form: only every `every_n`-th forward is instrumented and, when only_stage >= 0, only the two
 * events around that stage are recorded.  An event in the stream costs a ~5 us pipeline bubble (the next
 * kernel can no longer be launched under the previous one's tail), so this is what a throughput
 * measurement uses inside its timed region.  With only_stage = 1 (fused encoder kernel) stage 2 reads 0. */
int ita_profile_begin_sampled(ita_handle h, int max_forwards, int every_n, int only_stage);
int ita_profile_end(ita_handle h, double* stage_ms, int* n_forwards);

/* ---- fusion tail on large token grids (BASELINE config 5) ----------------
 * The layers of models/ITA_single_layer_upsample_shuffle/QAT/model.py:116-121 -- PixelShuffle(2) ||
 * Upsample(x2, bilinear, align_corners=True) -> cat -> Conv2d(5E/4 -> out_ch, 3, padding 1) -- on a
 * tok_h x tok_w token grid (models/ITA_upsample_shuffle/model.py:70-79 declares them with E = 128 and
 * 48 outputs):  x_dev (B, tok_h*tok_w, E) f32 -> out_dev (B, out_ch, 2 tok_h, 2 tok_w) f32.
 * ita_fusion_tail_load takes HOST pointers (conv weight (out_ch, 5E/4, 3, 3), bias (out_ch)), once;
 * it is independent of ita_load_weights.  Split-precision f16 MFMA, result within 1e-5 relative of
 * the f32 oracle.  Needs E % 16 == 0, out_ch <= 64, tok_h % 4 == 0, tok_w % 16 == 0. */
int ita_fusion_tail_load(ita_handle h, const float* conv_w_host, const float* conv_b_host, int E, int out_ch);
int ita_fusion_tail_large(ita_handle h, const float* x_dev, float* out_dev, int batch, int tok_h, int tok_w,
                          void* stream);

/* Diagnostic: one encoder layer (E = 64) with in-kernel s_memtime stamps, taken by waves 0 and 4 of every
 * workgroup over its first 8 frames, 16 slots each: stamps[((block * 8 + frame) * 2 + wave / 4) * 16 + slot],
 * a u64 device buffer of min(batch, #CUs) * 256 entries.  Slots: 0 frame start, 1 projections done, 2 past the
 * K/V^T barrier, 3 logits, 4 softmax, 5 A.V, 6 past the second barrier, 7 out_proj + LayerNorm1, 8 fc1,
 * 11 fc2 + LayerNorm2 + store; with image_u8_dev (tokenizer fused in front, x_dev unused) 9 and 10: patch blend
 * and conv + LayerNorm of the NEXT frame.  Not used by the product path. */
int ita_debug_encoder_stamps(ita_handle h, int layer, const float* x_dev, const void* image_u8_dev, float* y_dev,
                             int batch, unsigned long long* stamps_dev, void* stream);

/* Diagnostic / test entry: IntegerApproximatedSoftmax (models/ITA/QAT/ITA_softmax.py:51-61) alone, through the same
 * device function and register layout the encoder kernel uses between QK^T and A.V: int8 logits (rows,128) -> u8
 * probabilities (rows,128), both device pointers.  Needs no weights. */
int ita_debug_softmax_rows(ita_handle h, const int8_t* logits_dev, uint8_t* probs_dev, int rows, void* stream);

/* ---- wire format of the reference's UDP host (ita_wire.h), exported for bindings and tests ------- */
/* frame_out: [desired_velocity, position_x, quat w, x, y, z]; returns 0, or -1 on a short packet */
int ita_wire_unpack_packet(const uint8_t* packet, size_t nbytes, int quat_stride_bug, float* frame_out);
/* main.cpp:381-417: clip x, normalise, scale by desired velocity, near-start x override */
void ita_wire_postprocess(const float* raw3, float desired_velocity, float position_x, float* out3);

/* ---- drop-in symbols of the reference plugin ---------------------------------------------- */

/* Selects the context/layer the two `void` symbols below run, and the element type of their
 * host buffers (the reference's C side says uint16 'float16_t', ITA_dispatch.c:7, its MLIR side
 * says f32, ITA_spec.mlir:30-33; both are supported). */
int ita_bind_dispatch(ita_handle h, int layer, int dispatch_dtype);

/* Same symbol and prototype as ITA_dispatch.c:17-19.  input/output are HOST buffers of
 * 1 x 128 x E elements (E = 128 in ITA_spec.mlir); the body is the full int8 attention block
 * instead of the reference's 16 384-element copy.  Synchronous.  Errors: ita_last_error(). */
void ITASelfAttention_workgroup(const uint16_t* input, uint16_t* output);
/* The 10-argument expanded-memref form (ITA_dispatch.c:31-41): base/aligned/offset/size/stride x 2 */
void ITASelfAttention_workgroup_expanded(const uint16_t* binding0, const uint16_t* binding0_aligned,
                                         size_t binding0_offset, size_t binding0_size, size_t binding0_stride,
                                         uint16_t* binding1, uint16_t* binding1_aligned, size_t binding1_offset,
                                         size_t binding1_size, size_t binding1_stride);
/* FFN twin for the `abs` marker (models/ITA/export/ITA_ONNX.py:35-38), same buffer convention. */
void ITAFeedForward_workgroup(const uint16_t* input, uint16_t* output);

#ifdef __cplusplus
}
#endif
#endif /* ITA_MI355X_H_ */
