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

// A batch object carries its OWN copy of the table, taken when it is created (gpet_batch_create) and changed only through
// gpet_batch_set_option: every entry point that works on a batch installs that copy for the calling thread (OptionScope), so
// what a batch does never depends on what another thread sets while it runs.  Without a scope (context-level calls, batch
// creation before the copy exists) reads go to the process-wide table.
constexpr int kMaxOptions = 64;
struct OptionSet {
  int v[kMaxOptions];
};
void option_snapshot(OptionSet* out);  // the process-wide table as it is now
int option_set_in(OptionSet* s, const char* name, int value, int* previous);
int option_get_in(const OptionSet* s, const char* name, int* value);
struct OptionScope {  // installs `s` (may be null: no change) as the calling thread's table for the scope's lifetime
  explicit OptionScope(OptionSet* s);
  ~OptionScope();
  OptionScope(const OptionScope&) = delete;
  OptionScope& operator=(const OptionScope&) = delete;
  OptionSet* prev;
  bool active;
};
// the value of option `name` in the calling thread's table (the batch's copy inside an OptionScope, else the process-wide
// table: reads see later gpet_set_option calls).  An unknown name is a programming error and aborts.
int& option(const char* name);
int option_index(const char* name);  // (aborts on an unknown name) -- for call sites that look an option up on every launch
int& option_at(int index);
// gpet_set_option / gpet_get_option: 0 on success, -1 for an unknown name
int option_set(const char* name, int value, int* previous);
int option_get(const char* name, int* value);
int option_count();
const OptionDef& option_def(int i);

}  // namespace gpet
