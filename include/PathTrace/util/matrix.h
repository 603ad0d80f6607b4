// PathTrace/util/matrix.h -- part of the PathTrace API; the declarations live in PathTrace/detail/linear.h
#pragma once
#include <PathTrace/detail/linear.h>
