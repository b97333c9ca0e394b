// stand-in for the syntax-only check of integration/*.cpp (see ../../README.md)
#pragma once
#include <boost/filesystem.hpp>
