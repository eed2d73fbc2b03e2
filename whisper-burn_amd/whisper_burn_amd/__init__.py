"""whisper-burn_amd: MI355X-native Whisper hot path (mel frontend + encoder-decoder) behind the
whisper-burn tensor seams.  All compute lives in lib/libwhisper_hip.so (csrc/, gfx950 HIP);
this package is the thin host-side mirror of the reference interface."""
from .model import (WB_BF16, WB_F32, Session, Whisper, burn_record_tensors, decode_params, find_chunk_overlap, load_audio_waveform,  # noqa: F401
                    max_waveform_samples, pcm_s16_to_f32_dev, resample, resample_filter, wav_info,
                    prep_audio, stitch_windows, waveform_to_mels_dev, waveform_to_text, waveform_to_tokens,
                    window_extents)
from .tokens import SpecialTokens  # noqa: F401
from ._lib import WbError  # noqa: F401
