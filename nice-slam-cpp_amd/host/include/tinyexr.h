// Stand-in for the one tinyexr entry point the reference's reader calls (src/inputs/CoFusionReader.cpp:44: LoadEXR), same signature
// and result layout (RGBA floats, malloc'ed, width * height * 4).  A from-scratch reader for single-part scanline OpenEXR files with
// NONE / ZIPS / ZIP compression and HALF / FLOAT / UINT channels (nice-slam-cpp_amd/host/src/nsk_io.cpp); the reference's vendored
// tinyexr (deps/tinyexr, 10 kLoC) is not copied.  Channels map to R,G,B,A by name; a single channel (Y, Z, or anything else) is
// replicated into R,G,B.
#pragma once
#define TINYEXR_SUCCESS 0
#define TINYEXR_ERROR_CANT_OPEN_FILE (-7)
#define TINYEXR_ERROR_INVALID_DATA (-4)
#define TINYEXR_ERROR_UNSUPPORTED_FORMAT (-10)
int LoadEXR(float** out_rgba, int* width, int* height, const char* filename, const char** err);
void FreeEXRErrorMessage(const char* msg);
