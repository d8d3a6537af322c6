// Thread-local last-error string shared by the C-ABI translation units.
#pragma once
#include <string>
namespace parsy {
void set_last_error(const std::string& msg);
const std::string& last_error();
}  // namespace parsy
struct parsy_symbolic;
namespace parsy { struct Symbolic; }
const parsy::Symbolic* parsy_symbolic_cxx(const parsy_symbolic* s);
