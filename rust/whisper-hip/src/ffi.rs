//! Raw bindings of include/whisper_hip.h (plain pointers and sizes; nothing unwinds across the ABI).
//! Every function returns 0 or a negative `wb_status`; `wb_last_error()` holds the thread-local message.
//! tests/test_rust_shim.py checks names and argument counts of this block against the header.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

#[repr(C)]
pub struct wb_model {
    _private: [u8; 0],
}
#[repr(C)]
pub struct wb_session {
    _private: [u8; 0],
}
#[repr(C)]
pub struct wb_comm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug, PartialEq, Eq)]
pub struct wb_dims {
    pub n_mels: i32,
    pub n_audio_ctx: i32,
    pub n_audio_state: i32,
    pub n_audio_head: i32,
    pub n_audio_layer: i32,
    pub n_vocab: i32,
    pub n_text_ctx: i32,
    pub n_text_state: i32,
    pub n_text_head: i32,
    pub n_text_layer: i32,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct wb_decode_params {
    pub beam_size: i32,
    pub max_depth: i32,
    pub padding: i32,
    pub overlap_seconds: i32,
    pub max_n_offsets: i32,
    pub min_n_overlaps: i32,
    pub mask_until_len: i32,
    pub max_batch_windows: i32,
    pub tok_start_of_transcript: i32,
    pub tok_language: i32,
    pub tok_transcribe: i32,
    pub tok_no_timestamps: i32,
    pub tok_end_of_text: i32,
}

pub const WB_OK: c_int = 0;
pub const WB_ERR_ARG: c_int = -1;
pub const WB_ERR_SHAPE: c_int = -2;
pub const WB_ERR_IO: c_int = -3;
pub const WB_ERR_HIP: c_int = -4;
pub const WB_ERR_OOM: c_int = -5;
pub const WB_ERR_STATE: c_int = -6;
pub const WB_F32: c_int = 0;
pub const WB_BF16: c_int = 1; // retired: wb_model_load_* return an error

extern "C" {
    pub fn wb_model_load_dump_dir(dir: *const c_char, device: c_int, compute_dtype: c_int, out: *mut *mut wb_model) -> c_int;
    pub fn wb_model_load_burn_record(mpk_gz_path: *const c_char, cfg_path: *const c_char, device: c_int,
                                     compute_dtype: c_int, out: *mut *mut wb_model) -> c_int;
    pub fn wb_model_dims(m: *const wb_model, out: *mut wb_dims) -> c_int;
    pub fn wb_model_free(m: *mut wb_model);
    pub fn wb_model_set_frame_limit(m: *mut wb_model, whisper_geometry: c_int) -> c_int;
    pub fn wb_model_encoder_gemm(m: *const wb_model) -> c_int;
    pub fn wb_model_decoder_gemm(m: *const wb_model) -> c_int;
    pub fn wb_max_waveform_samples(n_frame_max: i64) -> i64;
    pub fn wb_prep_audio(device: c_int, pcm: *const c_float, n: i64, sample_rate: c_double, mel: *mut c_float,
                         n_frames: *mut i64) -> c_int;
    pub fn wb_forward_encoder(m: *mut wb_model, mel: *const c_float, b: c_int, t: c_int, out: *mut c_float) -> c_int;
    pub fn wb_forward_decoder(m: *mut wb_model, tokens: *const i32, n: c_int, l: c_int, enc: *const c_float, c: c_int,
                              logits: *mut c_float) -> c_int;
    pub fn wb_forward(m: *mut wb_model, mel: *const c_float, b: c_int, t: c_int, tokens: *const i32, l: c_int,
                      logits: *mut c_float) -> c_int;
    pub fn wb_decode_params_default(p: *mut wb_decode_params);
    pub fn wb_window_extents(n_samples: i64, sample_rate: c_int, window_len: i64, overlap_seconds: c_int,
                             starts: *mut i64, lens: *mut i64, cap: i64) -> i64;
    pub fn wb_waveform_to_tokens(m: *mut wb_model, pcm: *const c_float, n: i64, sample_rate: c_int,
                                 p: *const wb_decode_params, is_special: *const u8, win_begin: c_int, win_end: c_int,
                                 win_tokens: *mut i32, row_stride: i32, win_lens: *mut i32, stitched: *mut i32,
                                 stitched_cap: i64, n_stitched: *mut i64) -> c_int;
    pub fn wb_session_begin(m: *mut wb_model, pcm: *const c_float, n_pcm: i64, starts: *const i64, lens: *const i64,
                            n_windows: c_int, max_beams: c_int, padding: c_int, out: *mut *mut wb_session) -> c_int;
    pub fn wb_session_set_special_mask(s: *mut wb_session, is_special: *const u8) -> c_int;
    pub fn wb_session_step(s: *mut wb_session, new_tokens: *const i32, parent: *const i32, window: *const i32, n: c_int,
                           apply_special_mask: c_int, k: c_int, top_ids: *mut i32, top_logprobs: *mut c_float) -> c_int;
    pub fn wb_session_decode(s: *mut wb_session, p: *const wb_decode_params, out_tokens: *mut i32, row_stride: i32,
                             out_lens: *mut i32) -> c_int;
    pub fn wb_session_free(s: *mut wb_session);
    pub fn wb_wav_read_f32(path: *const c_char, out: *mut c_float, capacity: i64, n_samples: *mut i64) -> c_int;
    pub fn wb_wav_read_f32_any_rate(path: *const c_char, out: *mut c_float, capacity: i64, n_samples: *mut i64) -> c_int;
    pub fn wb_resample_len(n_in: i64, rate_in: i32, rate_out: i32) -> i64;
    pub fn wb_resample_dev(device: c_int, src_dev: *const c_float, n_in: i64, rate_in: i32, rate_out: i32,
                           dst_dev: *mut c_float, capacity: i64, n_out: *mut i64) -> c_int;
    pub fn wb_comm_unique_id(id128: *mut u8) -> c_int;
    pub fn wb_comm_init(id128: *const u8, rank: c_int, world: c_int, device: c_int, out: *mut *mut wb_comm) -> c_int;
    pub fn wb_comm_free(c: *mut wb_comm);
    pub fn wb_comm_allgather(comm: *mut c_void, send: *const c_void, recv: *mut c_void, bytes_per_rank: i64) -> c_int;
    pub fn wb_shard_partition(n_windows: i64, rank: c_int, world: c_int, lo: *mut i64, hi: *mut i64) -> c_int;
    pub fn wb_waveform_to_tokens_sharded(m: *mut wb_model, pcm: *const c_float, pcm_on_device: c_int, n: i64,
                                         sample_rate: c_int, p: *const wb_decode_params, is_special: *const u8,
                                         rank: c_int, world: c_int, allgather: wb_allgather_fn, user: *mut c_void,
                                         win_tokens: *mut i32, row_stride: i32, win_lens: *mut i32, win_cap: i64,
                                         stitched: *mut i32, stitched_cap: i64, n_stitched: *mut i64) -> c_int;
    pub fn wb_last_error() -> *const c_char;
    pub fn wb_version() -> *const c_char;
}

/// The exchange of the sharded path: `send` = this rank's bytes, `recv` = world x bytes_per_rank bytes in rank order.
pub type wb_allgather_fn = Option<unsafe extern "C" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void,
                                                       bytes_per_rank: i64) -> c_int>;

/// `wb_comm_allgather` under the callback's exact type (same symbol, `comm` as the user pointer).
pub unsafe extern "C" fn wb_comm_allgather_thunk(user: *mut c_void, send: *const c_void, recv: *mut c_void,
                                                  bytes_per_rank: i64) -> c_int {
    wb_comm_allgather(user, send, recv, bytes_per_rank)
}

/// Opaque user pointer type of callbacks (kept for completeness of the C vocabulary).
pub type wb_user = *mut c_void;
