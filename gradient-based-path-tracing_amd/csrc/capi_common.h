// capi_common.h — error plumbing shared by the C-ABI translation units.
// Reference convention: Error() throws fl_exception (src/flexception.h:8-24), uncaught -> terminate.
// Across the C ABI that becomes: non-zero status + message in gdpt_last_error().
#pragma once
#include <exception>
#include <stdexcept>
#include <string>

namespace gdpt {

void set_last_error(const std::string &msg);

// Test-only overrides (include/gdpt_debug.h): value of knob `name`, or `def` when no test has set it. The table is
// empty in every product run; no environment variable feeds it.
double debug_knob(const char *name, double def);
void debug_store_stamps(const unsigned long long *v, int n);   // diagnostic kernel's segment cycles -> gdpt_debug_get_stamps
inline int debug_knob_int(const char *name, int def) { return (int)debug_knob(name, (double)def); }

template <class F>
int guarded(F &&fn) {
    try {
        fn();
        set_last_error("");
        return 0;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return 1;
    } catch (...) {
        set_last_error("unknown error");
        return 1;
    }
}

} // namespace gdpt
