// stand-in for the syntax-only check of integration/*.cpp (see ../README.md)
#pragma once
#include <filesystem>
#include <fstream>
#include <string>
namespace boost { namespace filesystem {
	class path : public std::filesystem::path {
	public:
		path() {}
		path(const char *s) : std::filesystem::path(s) {}
		path(const std::string &s) : std::filesystem::path(s) {}
		path(const std::filesystem::path &p) : std::filesystem::path(p) {}
		std::string leaf() const { return filename().string(); }
		std::string file_string() const { return string(); }
		path parent_path() const { return path(std::filesystem::path::parent_path()); }
		path branch_path() const { return parent_path(); }
	};
	inline bool exists(const path &p) { return std::filesystem::exists(p); }
	inline path complete(const path &p) { return path(std::filesystem::absolute(p)); }
	inline path current_path() { return path(std::filesystem::current_path()); }
	inline std::string extension(const path &p) { return p.extension().string(); }
	inline unsigned long long file_size(const path &p) { return std::filesystem::file_size(p); }
	inline bool remove(const path &p) { return std::filesystem::remove(p); }
	typedef std::ifstream ifstream;
	typedef std::ofstream ofstream;
	typedef std::fstream fstream;
} }
