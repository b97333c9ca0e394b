// stand-in for the syntax-only check of integration/*.cpp (see ../../README.md)
#pragma once
#include <tuple>
namespace boost { using std::tuple; using std::make_tuple; using std::get; using std::tie; }
