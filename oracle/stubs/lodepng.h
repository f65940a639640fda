// Test-infrastructure stub for lodepng.h (not on disk). PNG decoding always
// fails; textures do not affect physics and the oracle models load none.
#ifndef ORACLE_STUB_LODEPNG_H_
#define ORACLE_STUB_LODEPNG_H_
#include <cstddef>
enum LodePNGColorType { LCT_GREY = 0, LCT_RGB = 2, LCT_PALETTE = 3, LCT_GREY_ALPHA = 4, LCT_RGBA = 6 };
struct LodePNGColorMode { LodePNGColorType colortype; unsigned bitdepth; };
struct LodePNGInfo { unsigned srgb_defined; };
namespace lodepng {
struct State { LodePNGColorMode info_raw{LCT_RGBA, 8}; LodePNGInfo info_png{0}; };
}
inline unsigned lodepng_decode(unsigned char** out, unsigned* w, unsigned* h, lodepng::State*,
                               const unsigned char*, size_t) {
  *out = nullptr; *w = 0; *h = 0; return 1;
}
inline const char* lodepng_error_text(unsigned) { return "PNG decoding unavailable in the oracle build"; }
inline size_t lodepng_get_raw_size(unsigned, unsigned, const LodePNGColorMode*) { return 0; }
#endif
