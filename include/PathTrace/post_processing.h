// PathTrace/post_processing.h -- tone mapping and gamma of the PathTrace API (host code, after rendering).
#ifndef PATHTRACE_POST_PROCESSING_H
#define PATHTRACE_POST_PROCESSING_H

#include <PathTrace/image/image.h>

// histogram-equalising tone map on a brightness heuristic; alpha is left alone
void toneMap(Image<> &image);
// scales rgb by brightness^(1/gamma - 1)
void gammaCorrect(Image<> &image, float gamma = 1.8F);
// toneMap followed by gammaCorrect
void postProcess(Image<> &image);

#endif
