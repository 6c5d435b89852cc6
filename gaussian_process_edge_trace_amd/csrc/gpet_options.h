// Process-wide tuning switches of the library: ONE table (gpet_options.hip) behind gpet_set_option / gpet_get_option
// (include/gpet_hip.h).  An option's initial value comes from the environment variable GPET_<NAME IN CAPITALS> if it is
// set, else from the table's default; INTEGRATION.md section 5 lists them with their meaning.
#pragma once

namespace gpet {

struct OptionDef {
  const char* name;
  int def, lo, hi;   // default and admissible range (values outside are clamped); lo == -1: -1 means "chosen automatically"
  const char* doc;
};

// the value of option `name` (a reference into the table: reads see later gpet_set_option calls).  An unknown name is a
// programming error and aborts.
int& option(const char* name);
// gpet_set_option / gpet_get_option: 0 on success, -1 for an unknown name
int option_set(const char* name, int value, int* previous);
int option_get(const char* name, int* value);
int option_count();
const OptionDef& option_def(int i);

}  // namespace gpet
