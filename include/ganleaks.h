/* ganleaks.h -- C ABI of libganleaks_hip.so (MI355X / gfx950).
 *
 * The reference (CarloSaccardi/GAN-Leaks) is pure Python and has no FFI of its own; the drop-in
 * boundary for its full-black-box attack path is the set of Python call signatures listed in
 * SURVEY.md 8(b).  Each entry point below names the reference interface it stands behind
 * (file:line relative to the reference tree).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative gl_status on failure and never throws;
 *     gl_last_error() returns a thread-local, NUL-terminated description of the last failure.
 *   - plain pointers and sizes only.  Pointers named *_dev are DEVICE pointers (hipMalloc'd by the
 *     caller, by gl_malloc, or by PyTorch-ROCm: tensor.data_ptr()); *_host are host pointers.
 *   - one gl_ctx per GPU per host thread; all work of a context is enqueued on its HIP stream
 *     (private by default, or the caller's via gl_ctx_set_stream) and is asynchronous unless the
 *     function says it synchronises.
 *   - image rows are C-contiguous [count][D] with D = C*H*W in the reference's NCHW order
 *     (attack_models/fbb.py:134-135).
 */
#ifndef GANLEAKS_H
#define GANLEAKS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GL_ABI_VERSION 1

typedef enum gl_status {
    GL_OK = 0,
    GL_ERR_INVALID = -1,   /* bad argument (NULL, negative size, unsupported shape) */
    GL_ERR_HIP = -2,       /* a HIP runtime call failed */
    GL_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    GL_ERR_STATE = -4,     /* object not ready (weights missing, bank not set) */
    GL_ERR_EMPTY_BANK = -5,/* bank shorter than one BATCH_SIZE: reference raises ValueError at fbb.py:83 */
    GL_ERR_RCCL = -6       /* librccl missing, or an RCCL call failed (gl_comm_*) */
} gl_status;

typedef struct gl_ctx gl_ctx;       /* opaque: device + stream + scratch */
typedef struct gl_dcgan gl_dcgan;   /* opaque: packed DCGAN / WGAN-GP generator */
typedef struct gl_lpips gl_lpips;   /* opaque: VGG16 + LPIPS v0.1 lin layers */
typedef struct gl_pggan gl_pggan;   /* opaque: packed progressive-GAN generator */
typedef struct gl_medgan gl_medgan; /* opaque: medGAN residual-MLP generator + autoencoder decoder */
typedef struct gl_comm gl_comm;     /* opaque: one rank of an RCCL communicator, bound to a gl_ctx */

/* ---------------------------------------------------------------- library / context */
int gl_abi_version(void);
const char *gl_last_error(void);
int gl_device_count(int *out_count);
int gl_ctx_create(int device, gl_ctx **out_ctx);           /* replaces `device = torch.device("cuda")`, fbb.py:40 */
int gl_ctx_destroy(gl_ctx *ctx);
int gl_ctx_set_stream(gl_ctx *ctx, void *hip_stream);      /* NULL restores the private stream */
int gl_ctx_get_stream(gl_ctx *ctx, void **out_hip_stream);
int gl_ctx_sync(gl_ctx *ctx);                              /* hipStreamSynchronize */

/* split-fp16 kernels store activations as fp16 halves of (value * 2^k); a value beyond the fp16 range is clamped and counted.
 * Returns (and resets) the number of workgroups that clamped since the last call; synchronises.  Non-zero means the result of
 * the split path is not trustworthy for these weights: rerun with gl_*_set_precision(.., 0). */
int gl_ctx_h3_saturations(gl_ctx *ctx, int64_t *out_count);

/* device memory + copies (synchronous w.r.t. the context stream) */
int gl_malloc(gl_ctx *ctx, size_t bytes, void **out_dev);
int gl_free(gl_ctx *ctx, void *dev);
/* gl_free keeps blocks of >= 256 MiB in the context (an arena) and gl_malloc hands them out again for requests of the same size (up to 1/8
 * smaller): the 153 GiB of query rows of a 256 x 256 attack cost 2-7 s to allocate and free per call otherwise.  gl_ctx_trim returns the kept
 * blocks to the driver (the library does it itself when one of its allocations runs out of memory; gl_ctx_destroy does it too). */
int gl_ctx_trim(gl_ctx *ctx);
/* device memory of the context's GPU: *out_available = bytes a gl_malloc could get right now (hipMemGetInfo's free bytes + the blocks the arena
 * keeps), *out_total = the device's memory.  attack() sizes the resident query rows of a streamed search from it. */
int gl_mem_info(gl_ctx *ctx, size_t *out_available, size_t *out_total);
int gl_memcpy_h2d(gl_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* `.to(device)`, fbb.py:135,141,145 */
int gl_memcpy_d2h(gl_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);   /* `.item()`, fbb.py:88 */
int gl_memset(gl_ctx *ctx, void *dev, int value, size_t bytes);

/* HIP events on the context stream (bench.py timing) */
int gl_event_create(void **out_event);
int gl_event_destroy(void *event);
int gl_event_record(gl_ctx *ctx, void *event);
int gl_event_elapsed_ms(void *start, void *stop, float *out_ms);   /* synchronises on `stop` */

/* per-kernel timing: while enabled, every launch of a tagged kernel is bracketed by two HIP events on
 * the context stream.  gl_prof_read synchronises and returns the summed duration and launch count
 * of one tag since the last gl_prof_reset. */
#define GL_PROF_GATHER_CONV 0   /* fp32-MFMA gather convolution (generator layers 0-3) */
#define GL_PROF_L2_KNN 1        /* int8-MFMA pairwise L2 + argmin */
#define GL_PROF_CONVT_RGB 2     /* generator tail: ConvT -> 3 channels + tanh + quantise */
#define GL_PROF_L2_PREPARE 3    /* u8 -> biased int8 + norms */
#define GL_PROF_FEAT_KNN 4      /* fp32-MFMA pairwise |V_q - V_n|^2 + argmin (l2-lpips) */
int gl_prof_enable(gl_ctx *ctx, int on);
int gl_prof_read(gl_ctx *ctx, int tag, double *out_total_ms, int64_t *out_launches);
int gl_prof_reset(gl_ctx *ctx);

/* ---------------------------------------------------------------- 8-bit image codec */
/* float images in [-1,1] -> u8 codes, exactly when x == fl32(2*(u/255.)-1)
 * (attack_models/utils.py:82 followed by fbb.py:134 `.float()`).  *off_lattice_dev (int32, device,
 * caller-zeroed) is incremented for every element that is not on that lattice. */
int gl_encode_lattice_f32(gl_ctx *ctx, const float *x_dev, int64_t count, uint8_t *u8_dev, int32_t *off_lattice_dev);
/* u8 -> float32 2*(u/255.)-1 (float64 arithmetic, utils.py:82) */
int gl_decode_u8(gl_ctx *ctx, const uint8_t *u8_dev, int64_t count, float *x_dev);
/* generator output -> u8 as the generate branches write it: mode 0 = Normalize(-1,2) then
 * ToPILImage mul(255).byte() (gan_models/dcgan/train_torch.py:154-158,172); mode 1 = x*0.5+0.5
 * (gan_models/pggan/train.py:238). */
int gl_quantize_f32(gl_ctx *ctx, const float *x_dev, int64_t count, int mode, uint8_t *u8_dev);

/* ---------------------------------------------------------------- L2 nearest neighbour (the hot path) */
/* bytes one prepared row occupies: D rounded up to the kernel's K tile */
int64_t gl_l2_row_stride(int64_t d);
/* u8 rows -> biased int8 rows (u-128, zero padded to gl_l2_row_stride(d)) and per-row sum (u-128)^2.
 * rows_i8_dev: [count][gl_l2_row_stride(d)] bytes; norms_dev: [count] int32 (the bit pattern of an unsigned value when d > 131071).
 * d <= 262143 (3 x 256 x 256 = 196608 fits). */
int gl_l2_prepare(gl_ctx *ctx, const uint8_t *rows_u8_dev, int64_t count, int64_t d, int8_t *rows_i8_dev, int32_t *norms_dev);
/* keys[q] = UINT64_MAX */
int gl_keys_init(gl_ctx *ctx, uint64_t *keys_dev, int64_t nq);
/* keys[q] = min(keys[q], (S(q,n) << shift) | (index_base + n)) over n in [0, n_rows):
 *   S = sum_k (uq_k - ub_k)^2, exact integers (d <= 66051: modulo 2^32 with S < 2^32; up to d = 262143: int32 segments of 64 KiB of K
 *   summed in 64 bits).  shift = 32 for d <= 33025 and one bit less per doubling of d beyond (31 at 3x128x128, 29 at 3x256x256) so that
 *   keys stay below 2^63 for the int64 all-reduce(min); index_base + n_rows <= 2^shift.  gl_keys_unpack derives the same shift from d.
 *   Replaces the loop body + torch.min of custom_knn
 *   (attack_models/fbb.py:77-86) with Loss('l2') (attack_models/utils.py:163,169,176) for the whole
 *   query set at once.  The caller applies the BATCH_SIZE truncation (fbb.py:77) by passing
 *   n_rows = n_eff.  index_base is the global index of bank row 0 (bank shards). */
int gl_l2_knn_i8(gl_ctx *ctx, const int8_t *bank_i8_dev, const int32_t *bank_norm_dev, int64_t n_rows, int64_t index_base,
                 const int8_t *query_i8_dev, const int32_t *query_norm_dev, int64_t nq, int64_t d, uint64_t *keys_dev);
/* keys -> (distance fp32 = fl32(S * 4/(255^2 d)), index int64).  `min_distance.item(), indices[min_index].item()`, fbb.py:88 */
int gl_keys_unpack(gl_ctx *ctx, const uint64_t *keys_dev, int64_t nq, int64_t d, float *dist_dev, int64_t *idx_dev);
/* The same exact path for tables of small non-negative integers (x == (float)u, u in 0..255: binary / count rows such as medGAN's thresholded
 * output, gan_models/medgan/train.py:306-312): encode to bytes, gl_l2_prepare + gl_l2_knn_i8 as for images, and
 * dist = fl32(S / d) = what an fp32 mean((y-x)^2) gives while its sum is exact (S < 2^24).  off_lattice_dev counts values that are not such integers. */
int gl_encode_integers_f32(gl_ctx *ctx, const float *x_dev, int64_t count, uint8_t *u8_dev, int32_t *off_lattice_dev);
int gl_decode_u8_integers(gl_ctx *ctx, const uint8_t *u8_dev, int64_t count, float *x_dev);
int gl_keys_unpack_integers(gl_ctx *ctx, const uint64_t *keys_dev, int64_t nq, int64_t d, float *dist_dev, int64_t *idx_dev);

/* out[i] = fl32(S(x_hat[i], x_gt[b_gt == 1 ? 0 : i]) * 4/(255^2 d)), i < b: the per-sample loss vector
 * Loss('l2').forward(x_hat, x_gt) returns (attack_models/utils.py:163,169,171-177; x_gt broadcasts
 * when it holds one image, fbb.py:79).  u8 rows on the device. */
int gl_l2_rows_u8(gl_ctx *ctx, const uint8_t *x_hat_u8_dev, int64_t b, const uint8_t *x_gt_u8_dev, int64_t b_gt, int64_t d, float *out_dev);

/* ---- arbitrary fp32 images (values off the 8-bit lattice): fixed-order fp32 evaluation of
 * mean((y - x)**2) (attack_models/utils.py:163): four interleaved fmaf chains over k, summed pairwise, divided by
 * d (exact definition in gan-leaks_amd/csrc/gl_l2f32.hip, shared bit for bit with the oracle).
 * keys[q] = min(keys[q], (float_bits(dist) << 32) | (index_base + n)); rows are [count][d] fp32, 16-byte aligned. */
int gl_l2_knn_f32(gl_ctx *ctx, const float *bank_dev, int64_t n_rows, int64_t index_base, const float *query_dev, int64_t nq, int64_t d,
                  uint64_t *keys_dev);
int gl_keys_unpack_f32(gl_ctx *ctx, const uint64_t *keys_dev, int64_t nq, float *dist_dev, int64_t *idx_dev);
/* Loss('l2').forward for fp32 inputs: out[i] = dist(x_hat[i], x_gt[b_gt == 1 ? 0 : i]) */
int gl_l2_rows_f32(gl_ctx *ctx, const float *x_hat_dev, int64_t b, const float *x_gt_dev, int64_t b_gt, int64_t d, float *out_dev);

/* One-call form for host callers: u8 images in HOST memory, results in HOST memory; applies the
 * truncation n_eff = (n_bank / batch_size) * batch_size itself.  Equals
 * [custom_knn(bank, q, Loss('l2'), args) for q in queries]  (attack_models/fbb.py:73-88,156-159).
 * Synchronises. Returns GL_ERR_EMPTY_BANK when n_bank < batch_size. */
int gl_fbb_knn_l2_host(gl_ctx *ctx, const uint8_t *bank_u8_host, int64_t n_bank, const uint8_t *queries_u8_host, int64_t nq,
                       int64_t d, int64_t batch_size, float *dist_host, int64_t *idx_host);

/* ---------------------------------------------------------------- sharded bank: the cross-GPU minimum (RCCL over xGMI) */
/* The reference runs on one device (attack_models/fbb.py:40) and takes the minimum over the whole bank with torch.min (fbb.py:86).  With the bank
 * sharded over GPUs every rank holds keys[q] = min over ITS rows, global indices inside; the minimum over ranks of the unsigned
 * 64-bit keys is the single-device result bit for bit (smallest distance, then smallest global index).  One communicator rank per gl_ctx;
 * librccl is bound at run time on the first gl_comm_* call (GL_ERR_RCCL if absent).
 *   one process per GPU : rank 0 calls gl_comm_unique_id, the launcher carries the GL_COMM_ID_BYTES bytes to every rank (any out-of-band
 *                         channel: torch.distributed's store, MPI, a file), every rank calls gl_comm_init_rank (collective, blocking);
 *   one process, N GPUs : gl_comm_init_all on N contexts of N different devices; a thread per context may then call gl_allreduce_min_keys,
 *                         or one thread issues all N calls between gl_comm_group_start / gl_comm_group_end. */
#define GL_COMM_ID_BYTES 128
int gl_comm_unique_id(void *id_out_host);                                                     /* ncclGetUniqueId */
int gl_comm_init_rank(gl_ctx *ctx, const void *id_host, int rank, int nranks, gl_comm **out);  /* ncclCommInitRank on ctx's device */
int gl_comm_init_all(gl_ctx *const *ctxs, int n, gl_comm **out_comms);                         /* ncclCommInitAll; out_comms[n] */
int gl_comm_destroy(gl_comm *comm);
/* ncclCommAbort: ends the communicator and whatever collective of it is still queued or running on ANY of its ranks' streams, without waiting.
 * For the error path of a multi-rank job: a rank that fails before its gl_allreduce_min_keys leaves the other ranks' reduce kernels waiting on the
 * GPU for ever; aborting every local communicator (from any host thread) lets their streams drain.  The handle is freed, as by gl_comm_destroy. */
int gl_comm_abort(gl_comm *comm);
int gl_comm_rank(const gl_comm *comm, int *out_rank, int *out_nranks);
/* keys_dev[q] = min over ranks of keys_dev[q], in place: ncclAllReduce(ncclMin, ncclUint64) queued on the context's stream -- behind the search
 * kernel that wrote the keys, ahead of gl_keys_unpack*; no host synchronisation.  Q x 8 bytes (80 KB at Q = 10^4): latency-bound. */
int gl_allreduce_min_keys(gl_comm *comm, uint64_t *keys_dev, int64_t nq);
/* recv_dev[r * bytes_per_rank ...] = rank r's send block, for every r: ncclAllGather on the context's stream (bytes as ncclUint8).  In place when
 * send_dev == recv_dev + rank * bytes_per_rank.  Used to shard the one part of the path that is replicated otherwise, the VGG16 features of the
 * queries (SURVEY.md 8e "shard it and all-gather if it shows up"): rank r featurises queries [r Q/N, (r+1) Q/N) and the search rows (1.02 MB
 * each at 64 x 64) are gathered; measured on one-rank shares, configs[2] projects to 5.8 x at 8 GPUs with replicated query features. */
int gl_allgather_rows(gl_comm *comm, const void *send_dev, void *recv_dev, int64_t bytes_per_rank);
int gl_comm_group_start(void);                                                                 /* ncclGroupStart */
int gl_comm_group_end(void);                                                                   /* ncclGroupEnd */

/* ---------------------------------------------------------------- DCGAN / WGAN-GP generator */
/* gan_models/dcgan/model_torch.py:75-96 == gan_models/wgangp/model.py:37-58:
 * 4 x [ConvTranspose2d(k4, bias=False) -> BatchNorm2d(eval) -> ReLU] (s1p0 then s2p1 x3),
 * ConvTranspose2d(k4,s2,p1)+bias -> tanh.  Output 64x64. */
int gl_dcgan_create(gl_ctx *ctx, int z_dim, int channels_img, int features_g, gl_dcgan **out);
int gl_dcgan_destroy(gl_dcgan *g);
/* weights in the reference's state_dict layouts, HOST pointers (PyTorch only reads the .pth):
 * layer 0..3: gen.{layer}.0.weight [C_in][C_out][4][4]; layer 4: gen.4.weight */
int gl_dcgan_set_conv_weight(gl_dcgan *g, int layer, const float *w_host);
/* gen.{layer}.1.{weight,bias,running_mean,running_var}, eps = 1e-5 (nn.BatchNorm2d default) */
int gl_dcgan_set_bn(gl_dcgan *g, int layer, const float *gamma_host, const float *beta_host, const float *mean_host,
                    const float *var_host, float eps);
int gl_dcgan_set_out_bias(gl_dcgan *g, const float *bias_host);   /* gen.4.bias */
/* z_dev [n][z_dim] fp32.  Either output may be NULL: out_f32_dev [n][C][64][64] (what
 * Generator.forward returns) and out_u8_dev [n][C][64][64] (the PNG bytes of the generate branch,
 * quantised as gl_quantize_f32 mode 0). */
int gl_dcgan_forward(gl_dcgan *g, const float *z_dev, int64_t n, float *out_f32_dev, uint8_t *out_u8_dev);
/* images per internal pass (activations for that many images stay resident); 0 = default */
int gl_dcgan_set_chunk(gl_dcgan *g, int64_t images_per_pass);
/* arithmetic of the ConvTranspose stack: 0 = fp32 MFMA (every product exact in fp32); 1 (default) = split-fp16: operands are
 * carried as hi + lo halves (~22 mantissa bits), three fp16 MFMAs per product, fp32 accumulation -- same error class,
 * 2-3x faster.  Stacks with the attention block (VAEGAN) always run mode 0. */
int gl_dcgan_set_precision(gl_dcgan *g, int mode);
/* 1 (default): in split-fp16 mode, when the last hidden layer has 64 or 128 channels (features_g = 32 or 64), the 3-channel output layer is computed in that
 * layer's epilogue and its 32 x 32 x C activations are never written to HBM (-9 % on the DCGAN-64 step); 0: separate launches.  Same values up to fp32
 * summation order. */
int gl_dcgan_set_fuse_tail(gl_dcgan *g, int on);
/* VAEGAN generator (gan_models/vaegan/train.py:109-135) = the same ConvTranspose stack with features_g = d/2, plus:
 * the epilogue of layers 0..3 set directly (the caller folds 1/sigma of SpectralNorm, the ConvTranspose bias and
 * BatchNorm into scale/shift), and SelfAttention (gan_models/vaegan/ops.py:86-120) on the 16 x 16 output of layer 2. */
int gl_dcgan_set_affine(gl_dcgan *g, int layer, const float *scale_host, const float *shift_host);
/* SpectralNorm(ConvTranspose2d) of layer 0..3 (gan_models/vaegan/ops.py:23-75): w_bar [C_in][C_out][4][4] (also installs the layer's weights), power-iteration
 * vectors u [C_in] and v [C_out * 16], bn_scale = gamma / sqrt(var + eps) and shift = (conv bias - mean) * bn_scale + beta per output channel.  Every
 * gl_dcgan_forward then advances u, v by `power_iterations` steps ON THE DEVICE and divides the layer's epilogue scale by sigma = u . W v, as the reference does
 * on every forward (also in eval mode).  gl_dcgan_get_spectral_state copies the current u, v back (state_dict). */
int gl_dcgan_set_spectral_norm(gl_dcgan *g, int layer, const float *w_bar_host, const float *u_host, const float *v_host, const float *bn_scale_host,
                               const float *shift_host, int power_iterations);
int gl_dcgan_get_spectral_state(gl_dcgan *g, int layer, float *u_host, float *v_host);
/* hold != 0: forwards reuse the current u, v, sigma (re-running the SAME call, e.g. in the other arithmetic mode) */
int gl_dcgan_set_spectral_hold(gl_dcgan *g, int hold);
int gl_dcgan_set_attention(gl_dcgan *g, const float *wq_host, const float *bq_host, const float *wk_host, const float *bk_host, const float *wv_host,
                           const float *bv_host, float gamma);

/* ---------------------------------------------------------------- PGGAN generator */
/* gan_models/pggan/model_torch.py:49-88 Generator(z_dim, in_channels, img_channels).forward(x, steps, alpha)
 * (WSConv2d :8-22, PixelNorm :25-31, ConvBlock :33-47).  in_channels must be a multiple of 32 and every block that
 * is used must keep >= 32 channels (steps <= 6 at in_channels = 512).  Weights are HOST pointers in the
 * reference's state_dict layouts. */
int gl_pggan_create(gl_ctx *ctx, int z_dim, int in_channels, int img_channels, gl_pggan **out);
int gl_pggan_destroy(gl_pggan *g);
/* initial.1.{weight [z][C][4][4], bias [C]}, initial.3.{conv.weight [C][C][3][3], bias [C]} */
int gl_pggan_set_initial(gl_pggan *g, const float *convt_w_host, const float *convt_b_host, const float *ws_w_host, const float *ws_b_host);
/* prog_blocks.{block}.conv1.{conv.weight, bias}, .conv2.{conv.weight, bias}; block 0..7 */
int gl_pggan_set_block(gl_pggan *g, int block, const float *conv1_w_host, const float *conv1_b_host, const float *conv2_w_host, const float *conv2_b_host);
/* rgb_layers.{j}.{conv.weight [nc][C_j][1][1], bias [nc]}; j = 0 is initial_rgb */
int gl_pggan_set_rgb(gl_pggan *g, int j, const float *w_host, const float *b_host);
int gl_pggan_set_chunk(gl_pggan *g, int64_t images_per_pass);
/* 1 (default) = split-fp16 convolutions (three fp16 MFMAs per product), 0 = fp32 MFMA */
int gl_pggan_set_precision(gl_pggan *g, int mode);
/* z_dev [n][z_dim] -> [n][nc][R][R], R = 4 * 2^steps.  out_f32_dev: what forward() returns; out_u8_dev: the bytes the
 * generate branch writes (gan_models/pggan/train.py:238-246: x*0.5+0.5, ToPILImage).  Either may be NULL. */
int gl_pggan_forward(gl_pggan *g, const float *z_dev, int64_t n, int steps, float alpha, float *out_f32_dev, uint8_t *out_u8_dev);

/* ---------------------------------------------------------------- medGAN (tabular) */
/* gan_models/medgan/model.py:44-73 Generator(z_dim, hidden_size) and :13-41 Autoencoder(input_size, hidden_size, binary).decode,
 * used together at gan_models/medgan/train.py:306-312.  z_dim == hidden_size == 128 (forced by the residual adds). */
int gl_medgan_create(gl_ctx *ctx, int z_dim, int hidden_size, int input_size, int binary, gl_medgan **out);
int gl_medgan_destroy(gl_medgan *g);
/* block 0/1 = gen_block{1,2}: .0.{weight [128][in], bias}, .1.{weight, bias, running_mean, running_var}; eps = 1e-3 (model.py:52,57) */
int gl_medgan_set_gen_block(gl_medgan *g, int block, const float *lin_w_host, const float *lin_b_host, const float *gamma_host, const float *beta_host,
                            const float *mean_host, const float *var_host, float eps);
/* decoder.0.{weight [input_size][hidden], bias [input_size]} */
int gl_medgan_set_decoder(gl_medgan *g, const float *w_host, const float *b_host);
/* Generator.forward: z_dev [n][128] -> hidden_out_dev [n][128] */
int gl_medgan_generate(gl_medgan *g, const float *z_dev, int64_t n, float *hidden_out_dev);
/* Autoencoder.decode: hidden_dev [n][128] -> decoded_dev [n][input_size]; binary_dev (may be NULL) = (decoded >= 0.5), train.py:311-312 */
int gl_medgan_decode(gl_medgan *g, const float *hidden_dev, int64_t n, float *decoded_dev, float *binary_dev);

/* ---------------------------------------------------------------- LPIPS (0.2 * LPIPS + L2, the reference's fbb distance) */
/* PerceptualLoss(model='net-lin', net='vgg') (attack_models/lpips_pytorch/__init__.py:9-32) -> PNetLin
 * (models/networks_basic.py:134-181) on torchvision's VGG16 `features` (models/pretrained_networks.py:96-134).
 * Features are computed once per image into a vector V with |V_q - V_n|^2 = 0.2*LPIPS(q,n) + mean((q-n)^2)
 * (attack_models/utils.py:176); see gan-leaks_amd/csrc/gl_lpips.hip. */
int gl_lpips_create(gl_ctx *ctx, gl_lpips **out);
int gl_lpips_destroy(gl_lpips *l);
/* conv_index 0..12 = torchvision vgg16().features convs {0,2,5,7,10,12,14,17,19,21,24,26,28}: weight [C_out][C_in][3][3], bias [C_out]; HOST pointers */
int gl_lpips_set_conv(gl_lpips *l, int conv_index, const float *w_host, const float *bias_host);
/* layer 0..4 = lin{layer}.model.1.weight of pretrained_models/v0.1/vgg.pth, [C] floats, all >= 0; HOST pointer */
int gl_lpips_set_lin(gl_lpips *l, int layer, const float *w_host);
int gl_lpips_set_chunk(gl_lpips *l, int64_t images_per_pass);
/* arithmetic of the 13 VGG16 convolutions: 1 (default) = split-fp16 (three fp16 MFMAs per product), 0 = fp32 MFMA */
int gl_lpips_set_precision(gl_lpips *l, int mode);
/* split-fp16 mode only: how activations are scaled into the fp16 halves.  1 (default) = one power of two per layer, chosen when the split path
 * first runs from a calibration pass of the fp32 pipeline over 16 fixed synthetic images (the same scales in every process and context for the
 * same weights; the layer's largest calibration activation lands in [1024, 2048) of the fp16 range 65504: 32-64 x headroom, full 22-bit
 * operands down to ~6e-5 of that maximum); 0 = the fixed factor 4 for every layer (rounds 1-2: fine while activations are O(1), saturates
 * beyond 16 376).  A store that still has to clamp is counted either way (gl_ctx_h3_saturations). */
int gl_lpips_set_calibration(gl_lpips *l, int enabled);
/* length of V for H x W images: sum_l C_l H_l W_l + 3 H W  (512 000 at 64 x 64); -1 if H or W is not a multiple of 16 */
int64_t gl_lpips_feature_dim(int H, int W);
/* images [n][3][H][W] (8-bit codes, or fp32 in [-1,1]) -> V_dev [n][K] feature rows of 4*K bytes each (opaque: every 32 values are
 * stored as 32 hi + 32 lo halves of V * 2^14, see csrc/gl_lpips.hip) and norms_dev [n] = |V|^2 */
int gl_lpips_features_u8(gl_lpips *l, const uint8_t *img_u8_dev, int64_t n, int H, int W, float *V_dev, float *norms_dev);
int gl_lpips_features_f32(gl_lpips *l, const float *img_f32_dev, int64_t n, int H, int W, float *V_dev, float *norms_dev);
/* keys[q] = min(keys[q], (float_bits(max(|V_q|^2 + |V_n|^2 - 2 V_q.V_n, 0)) << 32) | (index_base + n)), n < n_rows.
 * custom_knn (attack_models/fbb.py:73-88) with Loss('l2-lpips'); unpack with gl_keys_unpack_f32.  The contraction runs as three
 * fp16 MFMAs per product on the hi/lo halves (fp32 accumulation). */
int gl_feat_knn(gl_ctx *ctx, const float *bank_V_dev, const float *bank_norm_dev, int64_t n_rows, int64_t index_base, const float *query_V_dev,
                    const float *query_norm_dev, int64_t nq, int64_t K, uint64_t *keys_dev);
/* Search rows: the same V with ONE half per LPIPS value (V * 2^14 rounded once; measured effect on a distance <= 3e-7) and the image part kept
 * as hi/lo halves in three segments (query rows [hi|hi|lo], bank rows [hi|lo|hi], each padded to a multiple of 64), so that a plain fp16 dot of a
 * query row and a bank row is the split-fp16 product for the L2 term.  Row length gl_lpips_search_dim(H, W) halves (536 576 at 64 x 64 = 1.07 MB per
 * image instead of 2.05 MB).  role: 0 = query rows, 1 = bank rows.  norms_dev [n] = |row|^2 of the values the rows actually hold. */
int64_t gl_lpips_search_dim(int H, int W);
int gl_lpips_search_features_u8(gl_lpips *l, const uint8_t *img_u8_dev, int64_t n, int H, int W, int role, void *V16_dev, float *norms_dev);
int gl_lpips_search_features_f32(gl_lpips *l, const float *img_f32_dev, int64_t n, int H, int W, int role, void *V16_dev, float *norms_dev);
/* Lattice search rows, for 8-bit images on BOTH sides of the search (what fbb reads from PNG files, utils.py:60-84): a pixel 2 c / 255 - 1
 * is (2 c - 255) / 255, so with the row scale u = gl_lpips_lattice_scale(H, W) = 255 sqrt(3 H W) 2^e the image part of a row is the exact
 * fp16 integer (2 c - 255) 2^e: one K segment instead of the three of the hi / lo form, row length gl_lpips_lattice_dim(H, W) =
 * K_lpips + 3 H W halves (512 000 at 64 x 64), the L2 term exact up to the fp32 accumulation.  Queries and bank rows have the same
 * layout.  Search with gl_feat_knn_h1_scaled(..., K1 = gl_lpips_lattice_dim, row_scale = gl_lpips_lattice_scale). */
int64_t gl_lpips_lattice_dim(int H, int W);
float gl_lpips_lattice_scale(int H, int W);
/* MEMORY LAYOUT of fp16 search rows (both forms: gl_lpips_search_dim / gl_lpips_lattice_dim halves per row = K1), decided by K1 alone so that the
 * writers (gl_lpips_*_features_*) and the search (gl_feat_knn_h1*) agree without a flag:
 *   K1 * 2 <  2 MiB (images up to ~80 x 80) : row-major, V16_dev[row * K1 + k]
 *   K1 * 2 >= 2 MiB (96 x 96 and larger)   : K-blocked, half k of row r at byte ((r / 256) * (K1 / 64) + k / 64) * 32768 + (r % 256) * 128 + (k % 64) * 2
 *     -- the search reads the same 128-byte K slice of a tile's 256 + 256 rows together; with 16 MiB rows that is 512 different 2 MiB pages per slice
 *     (measured: 44 % of the TLB lookups missed, the kernel ran 17 % below its 64 x 64 rate); blocked, it is two contiguous 32 KiB pieces.
 * A buffer for n rows must hold gl_lpips_search_rows_capacity(n, K1) rows (n, or n rounded up to a multiple of 256), and V16_dev must point at a
 * multiple of 256 rows of its buffer when blocked.  Rows are opaque to callers either way (only |row|^2 and the keys come back). */
int64_t gl_lpips_search_rows_capacity(int64_t n, int64_t K1);
int gl_lpips_lattice_features_u8(gl_lpips *l, const uint8_t *img_u8_dev, int64_t n, int H, int W, void *V16_dev, float *norms_dev);

/* gl_feat_knn on search rows: same keys, one fp16 MFMA per product, 256 x 256 tiles.  K1 = gl_lpips_search_dim.
 * gl_feat_knn_h1_scaled: rows stored as V * row_scale (gl_feat_knn_h1: 2^14, the scale of gl_lpips_search_features_*). */
int gl_feat_knn_h1_scaled(gl_ctx *ctx, const void *bank_V16_dev, const float *bank_norm_dev, int64_t n_rows, int64_t index_base, const void *query_V16_dev,
                          const float *query_norm_dev, int64_t nq, int64_t K1, uint64_t *keys_dev, float row_scale);
int gl_feat_knn_h1(gl_ctx *ctx, const void *bank_V16_dev, const float *bank_norm_dev, int64_t n_rows, int64_t index_base, const void *query_V16_dev,
                   const float *query_norm_dev, int64_t nq, int64_t K1, uint64_t *keys_dev);
/* mean((y-x)^2) + argmin for ARBITRARY fp32 rows on the matrix cores (an alternative to the bit-reproducible VALU path gl_l2_knn_f32; ~15-60x
 * faster; distances agree to ~3e-6 * mean(x^2), i.e. ~1e-6 absolute for rows in [-1,1]): rows are stored as hi + lo halves of x * 2^e with a per-row power of two,
 * dist = |y|^2/d + |x|^2/d - 2 y.x/d with three fp16 MFMAs per product and fp32 accumulation.  V_dev: [n][gl_rows_split_dim(d)] 4-byte slots,
 * norms_dev [n] = |x|^2 / d, scales_dev [n] = 2^-e.  Keys as gl_l2_knn_f32 (unpack with gl_keys_unpack_f32). */
int64_t gl_rows_split_dim(int64_t d);
int gl_rows_split_f32(gl_ctx *ctx, const float *rows_f32_dev, int64_t n, int64_t d, void *V_dev, float *norms_dev, float *scales_dev);
int gl_rows_knn_split(gl_ctx *ctx, const void *bank_V_dev, const float *bank_norm_dev, const float *bank_scale_dev, int64_t n_rows, int64_t index_base,
                      const void *query_V_dev, const float *query_norm_dev, const float *query_scale_dev, int64_t nq, int64_t d, uint64_t *keys_dev);
/* custom_knn with Loss('l2-lpips') -- the reference's default fbb distance (attack_models/fbb.py:148) -- for host-resident 8-bit images in ONE call:
 * BATCH_SIZE truncation (fbb.py:77), search rows, the bank streamed through HBM so that about max_device_bytes (0 = 64 GiB) of feature rows are
 * resident at a time.  Returns GL_ERR_EMPTY_BANK where the reference raises at fbb.py:83. */
int gl_fbb_knn_lpips_host(gl_ctx *ctx, gl_lpips *l, const uint8_t *bank_u8_host, int64_t n_bank, const uint8_t *queries_u8_host, int64_t nq, int H, int W,
                          int64_t batch_size, int64_t max_device_bytes, float *dist_host, int64_t *idx_host);
/* Loss('l2-lpips').forward: per row, out_lpips = LPIPS and out_l2 = mean((y-x)^2) between V_hat[i] and V_gt[b_gt == 1 ? 0 : i];
 * K_lp = K - 3 H W is the length of the LPIPS part of V */
int gl_feat_rows_dist(gl_ctx *ctx, const float *V_hat_dev, int64_t b, const float *V_gt_dev, int64_t b_gt, int64_t K, int64_t K_lp, float *out_lpips_dev,
                      float *out_l2_dev);

#ifdef __cplusplus
}
#endif
#endif /* GANLEAKS_H */
