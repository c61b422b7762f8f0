/*
 * Callbacks.h — debug-hook C API of the host decoder (parseoggvorbis_amd/host).
 *
 * Interface-compatible with the reference's hook module (reference: src/Callbacks.h:43-79,88-94; semantics
 * src/Callbacks.cpp:224-370): same symbol names, argument meaning and on-disk dump format, so that the
 * reference's tests/compare-debug-out.py can parse what this decoder writes. The implementation (hooks.cpp)
 * is independent code.
 *
 * Threading contract (reference: src/Callbacks.h:16-21): registry calls are mutex-guarded; the set_data_*
 * selectors are thread_local and apply to the NEXT decoder registered on the calling thread; one decoder
 * lives on one thread.
 */
#ifndef PARSEOGGVORBIS_AMD_HOST_CALLBACKS_H_
#define PARSEOGGVORBIS_AMD_HOST_CALLBACKS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#include <string>
#include <vector>
extern "C" {
#endif

/* value type tags written into the dump (1 byte each) */
enum DataTypeId { DT_Float32 = 1, DT_Int32 = 2, DT_UInt32 = 3, DT_Uint8 = 4, DT_Bool = 5, DT_Int64 = 6, DT_UInt64 = 7 };

/* decoder registry: `ref` is any address identifying a decoder; aliases resolve to it */
void register_decoder_ref(const void* ref, const char* decoder_name, long sample_rate, int num_channels);
void register_decoder_alias(const void* orig_ref, const void* alias_ref);
void unregister_decoder_ref(const void* ref); /* harmless if unknown */

/* sink / filter for the next decoder registered on this thread */
void set_data_output_null(void);
void set_data_output_short_stdout(void);
void set_data_output_file(const char* fn);
void set_data_filter(const char** allowed_names); /* NULL-terminated list, or NULL for "everything" */

/* one named entry; channel < 0 = not per-channel; data == NULL = marker without payload */
void push_data_float(const void* ref, const char* name, int channel, const float* data, size_t len);
void push_data_u8(const void* ref, const char* name, int channel, const uint8_t* data, size_t len);
void push_data_i32(const void* ref, const char* name, int channel, const int32_t* data, size_t len);
void push_data_u32(const void* ref, const char* name, int channel, const uint32_t* data, size_t len);
void push_data_i64(const void* ref, const char* name, int channel, const int64_t* data, size_t len);
void push_data_u64(const void* ref, const char* name, int channel, const uint64_t* data, size_t len);
void push_data_int(const void* ref, const char* name, int channel, const int* data, size_t len);

const char* generic_itoa(uint32_t val, int base, int len);

#ifdef __cplusplus
}

void push_data_bool(const void* ref, const char* name, int channel, const std::vector<bool>& data);

/* true if the decoder behind `ref` has a sink that wants entries (lets the decoder skip device taps otherwise) */
bool decoder_wants_data(const void* ref);

struct ArgParser { /* --in ogg [--debug_out file] [--debug_stdout] [--help]  (reference: src/Callbacks.cpp:392-440) */
  std::string ogg_filename;
  void print_usage(const char* argv0);
  bool parse_args(int argc, const char** argv);
};
#endif

#endif
