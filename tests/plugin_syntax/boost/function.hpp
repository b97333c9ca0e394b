// stand-in for the syntax-only check of integration/*.cpp (see ../README.md)
#pragma once
#include <functional>
namespace boost { template <typename S> using function = std::function<S>; }
