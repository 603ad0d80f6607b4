// PathTrace/scene/light.h -- part of the PathTrace API; the declarations live in PathTrace/detail/world.h
#pragma once
#include <PathTrace/base.h>
#include <PathTrace/detail/world.h>
