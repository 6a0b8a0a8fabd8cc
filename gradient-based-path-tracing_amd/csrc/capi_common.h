// capi_common.h — error plumbing shared by the C-ABI translation units.
// Reference convention: Error() throws fl_exception (src/flexception.h:8-24), uncaught -> terminate.
// Across the C ABI that becomes: non-zero status + message in gdpt_last_error().
#pragma once
#include <exception>
#include <stdexcept>
#include <string>

namespace gdpt {

void set_last_error(const std::string &msg);

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
