// placeholder: stateful session entry points (filled in next)
#include "session.h"
using namespace wb;
extern "C" {
#define NOT_YET(name) do { wb::set_error(name ": not implemented yet"); return WB_ERR_STATE; } while (0)
int wb_session_begin(wb_model*, const float*, int64_t, const int64_t*, const int64_t*, int, int, int, wb_session**) { NOT_YET("wb_session_begin"); }
int wb_session_begin_mel(wb_model*, const float*, const int32_t*, int, int, int, wb_session**) { NOT_YET("wb_session_begin_mel"); }
int wb_session_set_special_mask(wb_session*, const uint8_t*) { NOT_YET("wb_session_set_special_mask"); }
int wb_session_step(wb_session*, const int32_t*, const int32_t*, const int32_t*, int, int, int, int32_t*, float*) { NOT_YET("wb_session_step"); }
int wb_session_last_logprobs(wb_session*, int, float*) { NOT_YET("wb_session_last_logprobs"); }
int wb_session_encoder_output(wb_session*, int, float*, int32_t*) { NOT_YET("wb_session_encoder_output"); }
void wb_session_free(wb_session*) {}
int wb_session_decode(wb_session*, const wb_decode_params*, int32_t*, int32_t, int32_t*) { NOT_YET("wb_session_decode"); }
int wb_waveform_to_tokens(wb_model*, const float*, int64_t, int, const wb_decode_params*, const uint8_t*, int, int, int32_t*, int32_t, int32_t*, int32_t*, int64_t, int64_t*) { NOT_YET("wb_waveform_to_tokens"); }
int wb_profile_enable(int) { return WB_OK; }
int wb_profile_read(double* o, int) { if (o) for (int i = 0; i < 8; i++) o[i] = 0; return WB_OK; }
}
