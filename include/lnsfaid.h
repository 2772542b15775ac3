/*
 * lnsfaid.h — C ABI of the MI355X-native batched LDPC decoder (50G-PON QC-LDPC, every value of the reference's
 * DecodeMethod switch: 0 NMS, 1 OMS, 2 LNS-FAID + DTBF, 3 OMS + BF, 4 OMS + DTBF, 5 LNS-FAID + 2B1C).
 *
 * This is the drop-in boundary for the reference's decoder member functions
 *   void CLDPC::Decode_OMS()        (reference CLDPC.h:148, CDecoder_OMS.cpp:13)
 *   void CLDPC::Decode_FAID()       (reference CLDPC.h:149, CDecoder_FAID.cpp:176)
 *   void CLDPC::Decode_FAID_2B1C()  (reference CLDPC.h:152, CDecoder_FAID_2B1C.cpp:96)
 *   void CLDPC::Decode()            (reference CLDPC.h:146, CLDPC.cpp:214; DecodeMethod 0 / default: normalised min-sum,
 *                                    Factor_1 / Factor_2 are numerators over 32, fixed iteration count, no early stop)
 *   int  CLDPC::Decode_OMSBF()      (reference CLDPC.h:150, CDecoder_OMSBF.cpp:13; DecodeMethod 3: the layered loop of
 *                                    Decode_OMS followed by plain bit flipping with threshold min(max vote, 5))
 *   int  CLDPC::Decode_OMS_DTBF()   (reference CLDPC.h:151, CDecoder_OMS_DTBF.cpp:18; DecodeMethod 4: the layered
 *                                    loop of Decode_OMS followed by the DTBF stage of Decode_FAID with other constants)
 *   Statistic CLDPC::CalculateErrors(...) (reference CLDPC.h:169, CLDPC.cpp:4819)
 * The reference has no FFI layer: inputs/outputs are the members `fixInput` /
 * `decodedBits` of `class CLDPC` (CLDPC.h:125-126) and the configuration is read
 * from Profile.txt (CTool.cpp:588-621) plus compile-time constants at the top of
 * each decoder file.  The entry points below carry exactly that information as
 * plain pointers and sizes.  INTEGRATION.md shows the CLDPC-side binding.
 *
 * Conventions: extern "C"; 0 = success, negative = error (lnsfaid_strerror);
 * no exceptions and no exit() across the boundary; the caller owns every host
 * buffer it passes, the context owns device buffers and its HIP stream.  One
 * context per (host thread, GPU): thread-compatible, not thread-safe.
 *
 * Batches are consecutive GROUPS of 32 codewords.  The group is part of the
 * contract: the reference decodes 32 codewords in lock-step and stops a group
 * only when all 32 lanes are clean (CDecoder_FAID.cpp:616, :6782), which is
 * observable in the hard decisions; this library reproduces it bit for bit.
 */
#ifndef LNSFAID_H
#define LNSFAID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LNSFAID_GROUP 32 /* codewords per group = Profile.txt noFrames = lanes of the reference's __m256i */

/* error codes */
#define LNSFAID_OK 0
#define LNSFAID_E_INVAL (-1)    /* bad argument / unsupported configuration          */
#define LNSFAID_E_CODE (-2)     /* H table is not the quasi-cyclic shape the kernels need */
#define LNSFAID_E_NOMEM (-3)    /* host or device allocation failed                  */
#define LNSFAID_E_HIP (-4)      /* a HIP runtime call failed (see lnsfaid_last_hip_error) */
#define LNSFAID_E_NODEVICE (-5) /* no usable GPU / extension built without device code */
#define LNSFAID_E_INTERNAL (-6)

/*
 * Code definition = the content of the reference's Constants_SSE.h
 * (Constants/50GPON-dc-original/Constants_SSE.h:4-19 and :29-3102).
 */
typedef struct lnsfaid_code {
    int32_t n_var;           /* _NoVar   17664 */
    int32_t n_check;         /* _NoCheck 3072  */
    int32_t n_edges;         /* _NoOnes  70400 */
    int32_t z;               /* Profile.txt Z  256 (circulant size) */
    int32_t puncture_tail;   /* number of tail VNs whose channel LLR is forced to 0
                                (CDecoder_FAID.cpp:253-255: 384) */
    int32_t nb_degres;       /* NB_DEGRES: number of consecutive row-degree classes */
    const int32_t* deg;      /* [nb_degres] DEG_k */
    const int32_t* deg_rows; /* [nb_degres] DEG_k_COMPUTATIONS */
    const uint16_t* pos_vn;  /* [n_edges] PosNoeudsVariable, row-major VN indices */
} lnsfaid_code;

/*
 * Decoder configuration = Profile.txt run-time keys + the compile-time
 * constants at the top of CDecoder_OMS.cpp / CDecoder_FAID.cpp /
 * CDecoder_FAID_2B1C.cpp.  lnsfaid_cfg_default() fills in the reference's
 * shipped values for a DecodeMethod.
 */
typedef struct lnsfaid_cfg {
    int32_t decode_method;     /* Profile.txt DecodeMethod 0..5 (README.md:13; 0 = NMS, CLDPC::Decode) */
    int32_t max_iteration;     /* Profile.txt MaxIteration (nb_iteration)           */
    int32_t factor_1;          /* Profile.txt Factor_1: OMS selective offset (>= 0), NMS numerator over 32 of min1 */
    int32_t factor_2;          /* Profile.txt Factor_2: OMS (>= 1: smaller values leave the 3-bit alphabet, E_INVAL), NMS numerator of min2 */
    int32_t floor_err_count;   /* CDecoder_OMS.cpp:28 (100) / FAID :193 (0) / 2B1C :117 (50) */
    int32_t floor_iter_thresh; /* CDecoder_OMS.cpp:29 (4)  / FAID :194 (-1) / 2B1C :118 (6) */
    int32_t ef_elimination;    /* EF_ELIMINATION 0 (FAID :6), 1 (2B1C :5) or, DecodeMethod 2 only, 2 (FAID :673-680) */
    int32_t max_bf_iter;       /* _maxBFiter 10 (CDecoder_FAID.cpp:208); 0 for OMS   */
    int32_t bf_L0;             /* _L0 50 (FAID :168) / 100 (2B1C :88)                */
    int32_t bf_L1;             /* _L1 0                                              */
    int32_t bf_alpha;          /* _alpha 1                                           */
    int32_t bf_delta;          /* _delta 1                                           */
    int32_t regular_col_weight;/* REGULAR_COL_WEIGHT 3 (CTool.h:6)                   */
    int32_t hard2_threshold;   /* 13 (CDecoder_FAID_2B1C.cpp:6130)                   */
    int32_t bf_vote_cap;       /* 5: plain BF flips votes >= min(max_vote, 5) (CDecoder_OMSBF.cpp:3332) */
    /* V2C_map_it{1..6}_[weight class 3,6,11,other][min(|t|,7)] (CDecoder_FAID.cpp:12-49) */
    int8_t v2c_map[6][4][8];
    /* V2C_map_it{1..6}_ef (CDecoder_FAID.cpp:130-165) */
    int8_t v2c_map_ef[6][4][8];
} lnsfaid_cfg;

/* per-group execution record (used for the algorithmic-byte accounting) */
typedef struct lnsfaid_group_stats {
    int32_t iterations;    /* I: layered iterations executed by the group            */
    int32_t bf_iterations; /* J: bit-flipping iterations that reached the flip step  */
} lnsfaid_group_stats;

typedef struct lnsfaid_ctx lnsfaid_ctx;

/* ---- code / configuration helpers (host only, no GPU needed) ---------------- */

/* Number of edges / VNs / checks of the built-in 50G-PON mother code and its
 * expansion into the Constants_SSE.h table format.  `pos_vn` must hold 70400
 * entries.  Returns LNSFAID_OK. */
int lnsfaid_code_50gpon(lnsfaid_code* code, uint16_t* pos_vn, int32_t* deg3, int32_t* deg_rows3);

/* Fill `cfg` with the reference's shipped constants for DecodeMethod 0..5. */
int lnsfaid_cfg_default(lnsfaid_cfg* cfg, int32_t decode_method, int32_t max_iteration);

/* Replace cfg->v2c_map by one of the table sets the reference selects at compile time in CDecoder_FAID.cpp
 * (#define FAID3 - the shipped default -, FAID32 or FAID2, CDecoder_FAID.cpp:8, :12-127). */
#define LNSFAID_TABLES_FAID3 0
#define LNSFAID_TABLES_FAID32 1
#define LNSFAID_TABLES_FAID2 2
int lnsfaid_cfg_table_preset(lnsfaid_cfg* cfg, int32_t preset);

/* The reference's compile-time switch EF_ELIMINATION of Decode_FAID (CDecoder_FAID.cpp:6, :192-203) for a DecodeMethod 2
 * configuration: 0 = off (the shipped build: floor_err_count 0, floor_iter_thresh -1), 1 = error-floor tables V2C_map_it*_ef
 * on unsatisfied rows inside the window (100, 6), 2 = the same plus the erasure of CDecoder_FAID.cpp:673-680 (20, 6): inside the
 * window the V2C message of a variable node of column weight REGULAR_COL_WEIGHT all of whose checks were unsatisfied at the
 * iteration's syndrome stage is set to 0, once per iteration.  Modes 1 and 2 need tables that are uniform over the weight classes
 * and non-decreasing (lnsfaid_create / lnsfaid_set_cfg return LNSFAID_E_INVAL otherwise). */
int lnsfaid_cfg_ef_elimination(lnsfaid_cfg* cfg, int32_t mode);

/* ---- decoder context --------------------------------------------------------- */

/* Replaces CLDPC::Initial (CLDPC.cpp:4772-4817): validates the code (must be
 * quasi-cyclic with circulant size z and no repeated block column inside a
 * block row), uploads the tables and allocates device state for up to
 * `max_groups` groups on GPU `device`. */
int lnsfaid_create(lnsfaid_ctx** ctx, const lnsfaid_code* code, const lnsfaid_cfg* cfg,
                   int32_t device, size_t max_groups);
void lnsfaid_destroy(lnsfaid_ctx* ctx);

/* Change the run-time configuration (the reference re-reads Profile.txt on
 * every Decode_* call, CDecoder_FAID.cpp:178-179). */
int lnsfaid_set_cfg(lnsfaid_ctx* ctx, const lnsfaid_cfg* cfg);

/* ---- the hot path ------------------------------------------------------------ */

/*
 * Replaces Decode_OMS / Decode_FAID / Decode_FAID_2B1C for n_groups groups.
 *   fixInput    host, int8 in [-7,7], per group the reference layout
 *               [32][K] information LLRs followed by [32][M] parity LLRs
 *               (CLDPC.h:126, CDecoder_FAID.cpp:217-241); group g starts at
 *               g * 32 * n_var.
 *   decodedBits host, int8 0/1, per group [32][n_var] (CLDPC.h:125,
 *               CDecoder_FAID.cpp:7102); group g starts at g * 32 * n_var.
 *   stats       optional, [n_groups].
 */
int lnsfaid_decode(lnsfaid_ctx* ctx, const int8_t* fixInput, size_t n_groups,
                   int8_t* decodedBits, lnsfaid_group_stats* stats);
/* The reference's call shape - one group per call from T worker threads, each with its own CLDPC / context (reference
 * CSimulate.cpp:136-164, main.cpp:164-172) - is served by a call combiner: the concurrent lnsfaid_decode calls of contexts
 * created with max_groups == 1 on the same device and for the same code are decoded in common launches (from four such
 * contexts on; results bit-identical to separate launches; the call still returns only when its own group is done).
 * Environment: LNSFAID_COALESCE=0 switches it off, LNSFAID_COMB_WORKERS (1..4, default 2) sets the batches in flight,
 * LNSFAID_COMB_BATCHES (default: the number of workers) into how many batches the members' calls are cut,
 * LNSFAID_SYNC=spin|block overrides how the host waits for the GPU (default: sleep from five live contexts on). */

/* Same with device-resident buffers (pointers valid on the context's GPU).  Work is queued on the context's stream; the
 * call synchronises with it once per kernel launch (it reads back how many codewords are still open: 1 launch when
 * nothing converges, typically 3 with early stop) and returns after the batch is complete.  d_stats optional. */
int lnsfaid_decode_device(lnsfaid_ctx* ctx, const int8_t* d_fixInput, size_t n_groups,
                          int8_t* d_decodedBits, lnsfaid_group_stats* d_stats);

/*
 * Replaces CLDPC::CalculateErrors (CLDPC.cpp:4842-4876) for n_groups groups:
 * compares the first K bits of every decoded frame with inputBits
 * ([32][K] per group, int8 0/1) and ADDS to
 *   out[0] TestFrame, out[1] ErrorFrame, out[2] ErrorBits, out[3] LT3ErrBitFrame.
 * inputBits == NULL means the all-zero codeword (FakeEncoder with the shipped
 * CodeWord_sym, CLDPC.cpp:163 / Codeword.h:4).
 */
int lnsfaid_count_errors(lnsfaid_ctx* ctx, const int8_t* decodedBits, const int8_t* inputBits,
                         size_t n_groups, uint64_t out[4]);
int lnsfaid_count_errors_device(lnsfaid_ctx* ctx, const int8_t* d_decodedBits,
                                const int8_t* d_inputBits, size_t n_groups, uint64_t out[4]);

/* ---- front-end on the device (optional, the host generator stays the parity source) ---- */

/*
 * Generates the fixInput of n_streams groups on the GPU, one group per reference worker thread ("stream"):
 * replaces, for one pass of the loop body of CSimulate::Run (CSimulate.cpp:126-132),
 *   CChannel::AWGNChannel (CChannel.cpp:71-97, Wichmann-Hill + Box-Muller, seed table CSimulate.cpp:11-17),
 *   CModulate::Demodulation + AfterDeModulationDeInterleaver (CModulate.cpp:152-212, :273-293) and
 *   CLDPC::float2LimitChar_4bit (CLDPC.cpp:4553-4573).
 *   seeds[s]        RandomSeed of stream s (IX = IY = IZ = seed, CChannel.cpp:121)
 *   draws_before[s] uniforms stream s has consumed so far; one group consumes lnsfaid_frontend_draws_per_group()
 *   mod_type        Profile.txt modType: 2 (QPSK), 4, 6, 8 (16-, 64-, 256-QAM); InterleaveModType: lnsfaid_frontend_set_interleave
 *   sigma           CSimulate::Configure's sigma (CSimulate.cpp:69-74); the channel adds N(0, (sigma/sqrt 2)^2) per axis
 *   codeword        host, [n_var] bits 0/1 sent in every frame (FakeEncoder), NULL = all-zero
 *   d_fixInput      device, n_streams groups in the decoder's layout
 * Integer and float stages are bit-exact; Box-Muller uses the device's double log / cos (directly, or as the arbiter of the
 * single-precision fast path, see lnsfaid_frontend_set_exact), so single LLRs can differ from the host generator's (rate
 * bounded in tests/test_gpu_frontend.py).
 */
int lnsfaid_frontend_device(lnsfaid_ctx* ctx, const uint32_t* seeds, const uint64_t* draws_before, size_t n_streams,
                            int32_t mod_type, float sigma, float scale, const int8_t* codeword, int8_t* d_fixInput);
/* Same with an explicit generator state per stream: states[3 s .. 3 s + 2] = RS.IX, RS.IY, RS.IZ of stream s at
 * draws_before[s] = 0, e.g. a row of the lastSeed table the driver writes to Temp.txt (main.cpp:200-207), which the
 * reference compiles back in under CONTINUE_SEED (CChannel.cpp:4-41, :116-119). */
int lnsfaid_frontend_device_states(lnsfaid_ctx* ctx, const uint32_t* states, const uint64_t* draws_before, size_t n_streams,
                                   int32_t mod_type, float sigma, float scale, const int8_t* codeword, int8_t* d_fixInput);
uint64_t lnsfaid_frontend_draws_per_group(const lnsfaid_ctx* ctx, int32_t mod_type);

/* The front-end kernel takes a quantised LLR from a single-precision evaluation of Box-Muller whenever no quantiser threshold
 * lies within that evaluation's error bound, and recomputes the symbol in double precision otherwise (a few in ten thousand):
 * same output, about a quarter of the time.  lnsfaid_frontend_set_exact(ctx, 1) (or LNSFAID_FRONTEND_EXACT=1 at creation)
 * sends every symbol through the double-precision chain, written with the reference's own integer generator and float
 * divisions - the A/B reference of the tests.
 * lnsfaid_frontend_fastpath_bounds scans EVERY float u in [0, 1) on the device: measured[0] = max |sqrt(-2 ln(1 - u)) - fast|,
 * measured[1] = max |cos(2 pi u) - fast|; assumed[] = what the kernel's bound uses.  measured must stay below assumed (the GPU
 * test asserts a factor of two): run it once on a device whose transcendental units are not those of gfx950. */
int lnsfaid_frontend_set_exact(lnsfaid_ctx* ctx, int32_t exact);
int lnsfaid_frontend_fastpath_bounds(lnsfaid_ctx* ctx, double measured[2], double assumed[2]);

/* Profile.txt InterleaveModType for lnsfaid_frontend_device: the block interleaver of BeforeModulationInterleaver /
 * AfterDeModulationDeInterleaver (CModulate.cpp:95-212) inside every frame; 1 (the default and the shipped value) is the
 * identity.  Must divide n_var. */
int lnsfaid_frontend_set_interleave(lnsfaid_ctx* ctx, int32_t interleave_mod_type);

/* Frames for lnsfaid_frontend_device when every stream sends its own 32 frames (the reference with a real encoder:
 * GenMsgSeq + Encode once per 50 calls, CSimulate.cpp:106-116) instead of one codeword in every frame.
 *   outputBits  host, n_streams groups in CLDPC::Encode's output layout ([32][K] then [32][M] per group)
 *   inputBits   host, their information bits [n_streams][32][K] (what CalculateErrors compares with)
 * Both are copied to the device and used by every following lnsfaid_frontend_device call that passes codeword = NULL and
 * at most n_streams streams; outputBits = NULL switches back.  lnsfaid_frontend_input_bits returns the device copy of
 * inputBits for lnsfaid_count_errors_device (NULL while no frames are set). */
int lnsfaid_frontend_set_frames(lnsfaid_ctx* ctx, const int8_t* outputBits, const int8_t* inputBits, size_t n_streams);
int lnsfaid_frontend_input_bits(lnsfaid_ctx* ctx, const int8_t** d_inputBits);

/* The context's own device staging buffers (each max_groups * 32 * n_var bytes): the fixInput buffer the
 * host-pointer entry points copy into and the decodedBits buffer they copy out of.  A host driver without its own
 * device allocator (host/CLDPC.cpp) runs front-end -> decode -> counters on them with the *_device entry points. */
int lnsfaid_io_buffers(lnsfaid_ctx* ctx, int8_t** d_fixInput, int8_t** d_decodedBits, lnsfaid_group_stats** d_stats);

/* Page-lock / release a host buffer (hipHostRegister / hipHostUnregister) for callers that have no HIP toolchain of
 * their own: with fixInput and decodedBits both pinned, lnsfaid_decode overlaps its copies with the decode (pieces of
 * whole groups on separate streams); pageable buffers are copied, decoded and copied back in sequence.  Unregister before
 * the buffer is freed. */
int lnsfaid_host_register(void* ptr, size_t bytes);
int lnsfaid_host_unregister(void* ptr);

/* Copy the per-group iteration counts a *_device decode left in the context's statistics buffer (the d_stats of
 * lnsfaid_io_buffers) to the host: what Decode_OMSBF / Decode_OMS_DTBF return as BFiter (reference CLDPC.h:150-151,
 * histogrammed into iterCount.txt by CSimulate.cpp:148-178) when the decoded frames themselves stay on the device. */
int lnsfaid_read_stats(lnsfaid_ctx* ctx, lnsfaid_group_stats* stats, size_t n_groups);

/* ---- multi-GPU: counters summed over RCCL (reference main.cpp:174-182) ---------------------
 * Groups of 32 codewords are independent, so a batch shards over GPUs as contiguous ranges of whole groups with no data-path
 * traffic; the only exchange is the sum of {TestFrame, ErrorFrame, ErrorBits, LT3ErrBitFrame} the reference's main thread
 * forms after pthread_join.  One context per GPU (one process or one host thread each):
 *   rank 0:     lnsfaid_comm_unique_id(id), hand `id` to the other ranks by any means (file, pipe, MPI, torch store)
 *   every rank: lnsfaid_comm_init(ctx, n_ranks, rank, id)       (collective: returns when all ranks have called it)
 *   per round:  lnsfaid_allreduce_counters(ctx, counters)       (in place: every rank ends up with the sums;
 *                                                                one ncclAllReduce of 4 x uint64 on the context's stream)
 * lnsfaid_comm_attach takes an existing ncclComm_t of the caller (not destroyed by the library) instead of creating one.
 * RCCL is looked up at run time (the copy already loaded in the process, else librccl.so.1): LNSFAID_E_NODEVICE if absent. */
#define LNSFAID_COMM_ID_BYTES 128
int lnsfaid_comm_unique_id(uint8_t id[LNSFAID_COMM_ID_BYTES]);
int lnsfaid_comm_init(lnsfaid_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t id[LNSFAID_COMM_ID_BYTES]);
int lnsfaid_comm_attach(lnsfaid_ctx* ctx, void* nccl_comm);
int lnsfaid_comm_destroy(lnsfaid_ctx* ctx);
int lnsfaid_allreduce_counters(lnsfaid_ctx* ctx, uint64_t counters[4]);

/* Which decode kernel the context launches.  rows_per_lane 0 (default): chosen per configuration - the byte-parallel kernel
 * with four check rows per lane and one wavefront per codeword for DecodeMethods 1..5 with FAID tables that are uniform over
 * the weight classes and non-decreasing (every set the reference ships) and for DecodeMethod 0 with Factor_1 == Factor_2 in
 * 15 .. 2114 (one normalisation factor whose scaled minimum has at most 16 levels), the two-rows-per-lane kernel otherwise
 * (NMS with two factors or a factor outside that range, other tables); 2 / 4 force one of them (4: LNSFAID_E_INVAL where it does
 * not apply).  Both produce identical results; the
 * switch exists for tests and A/B timing.  lnsfaid_kernel_rows_per_lane returns what the next decode will launch. */
int lnsfaid_select_kernel(lnsfaid_ctx* ctx, int32_t rows_per_lane);
int lnsfaid_kernel_rows_per_lane(const lnsfaid_ctx* ctx);

/* EXPERIMENTAL.  Wavefronts per codeword of the four-rows-per-lane kernel: 1 (default; 0 selects the default) or 2
 * (lnsfaid_kernel5.hip: the edges of a layer are dealt to two waves, four waves per SIMD instead of two; DecodeMethods 1..5
 * without the erasing EF_ELIMINATION 2, messages streamed through HBM).  Results are identical; LNSFAID_E_INVAL where it does
 * not apply.  Environment: LNSFAID_WAVES_PER_CODEWORD=2 forces it for every context it applies to.  lnsfaid_kernel_waves
 * returns what the next decode will launch. */
int lnsfaid_select_waves(lnsfaid_ctx* ctx, int32_t waves_per_codeword);
int lnsfaid_kernel_waves(const lnsfaid_ctx* ctx);

/* Where the four-rows-per-lane kernel keeps the check-to-variable messages (the reference's var_msgs, CLDPC.h:123,
 * lifetime CDecoder_FAID.cpp:211-214 ... :923) between the layers of a launch.  LNSFAID_MSG_REGISTERS: the compressed messages
 * of the codeword (72 dwords per lane for the 12 layers of the 50G-PON code) stay in the wavefront's registers for the whole
 * launch and reach HBM only when a codeword parks - no vector-memory operation inside the layer loop; available for codes of
 * up to 12 layers, not for EF_ELIMINATION 2 and not for DecodeMethod 0.  LNSFAID_MSG_HBM: streamed through HBM one layer ahead of use (every code).
 * 0 (default): registers where available.  Identical results either way; the switch exists for tests and A/B timing.
 * lnsfaid_message_store returns what the next decode will use. */
#define LNSFAID_MSG_REGISTERS 1
#define LNSFAID_MSG_HBM 2
int lnsfaid_select_message_store(lnsfaid_ctx* ctx, int32_t where);
int lnsfaid_message_store(const lnsfaid_ctx* ctx);

/* Workgroups (codewords) of the selected decode kernel a compute unit holds at once, from the HIP occupancy query, next to
 * what the kernel's LDS footprint alone would allow (50G-PON: 8 and 8).  A smaller first number means a build lost residency to
 * registers - about 40 % of the throughput for the one-wave-per-codeword kernel.  Also checks that the kernel has no static LDS
 * (LNSFAID_E_INTERNAL otherwise; every decode call checks the same once per kernel instance). */
int lnsfaid_kernel_residency(lnsfaid_ctx* ctx, int32_t* workgroups_per_cu, int32_t* lds_limit);

/* ---- measurement hooks ------------------------------------------------------- */

/* Device time (HIP events on the context's stream) and launch count of the
 * decode kernel accumulated since the last reset: out_ms = total kernel
 * milliseconds, out_launches = number of kernel launches (launches queued ahead
 * on a batch that turned out to be complete already are counted too: microseconds
 * each).  Calls of a one-group context that went through the call combiner are
 * not in it (they ran in launches shared with other contexts). */
int lnsfaid_kernel_time(lnsfaid_ctx* ctx, double* out_ms, uint64_t* out_launches, int32_t reset);

/* The HIP stream of the context as an opaque pointer (hipStream_t). */
void* lnsfaid_stream(lnsfaid_ctx* ctx);

const char* lnsfaid_strerror(int err);
const char* lnsfaid_last_hip_error(void);
const char* lnsfaid_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LNSFAID_H */
