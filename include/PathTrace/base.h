// PathTrace/base.h -- part of the PathTrace API; declarations in PathTrace/detail/core.h
#pragma once
#include <PathTrace/util/vector.h>
#include <PathTrace/util/matrix.h>
#include <PathTrace/detail/core.h>
