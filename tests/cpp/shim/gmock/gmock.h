// see ../gtest/gtest.h: the shim provides the matchers as well
#pragma once
#include <gtest/gtest.h>
