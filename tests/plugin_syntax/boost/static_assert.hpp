// stand-in for the syntax-only check of integration/*.cpp (see ../README.md)
#pragma once
#define BOOST_STATIC_ASSERT(x) static_assert(x, #x)
