// FQEntry.h -- one FASTQ record as four views into its batch, plus the framing checks.
// Role of reference src/FQEntry.{h,cpp}; the checks and their messages (src/FQEntry.cpp:53-97)
// are part of the CLI's observable behaviour and are reproduced.
#ifndef SICKLE_FQENTRY_H
#define SICKLE_FQENTRY_H

#include <cstdlib>
#include <string_view>
#include <type_traits>

#include "GZReader.h"

class FQEntry {
public:
    FQEntry() : position(0) {}
    // takes the next four lines of the batch; `previous` = position of the record before it
    FQEntry(int previous, Batch *reader);
    // the record at lines first_line..first_line+3 of the batch, NOT validated (parallel framing)
    FQEntry(const Batch &batch, size_t first_line, int position_);
    bool well_formed() const; // the checks of validate(), silently
    std::string_view name;
    std::string_view comment;
    std::string_view seq;
    std::string_view qual;
    int position; // 1-based record number, only used in error messages
    void validate() const; // exit(EXIT_FAILURE) with the reference's messages on a malformed record
};

// An array of trivially copyable records whose elements are NOT initialised on resize: every
// element is written exactly once by whoever fills it (in parallel), and value-initialising a
// hundred megabytes of them first on one thread was a fifth of the framing stage.
template <typename T> class RawVec {
public:
    RawVec() = default;
    RawVec(const RawVec &) = delete;
    RawVec &operator=(const RawVec &) = delete;
    ~RawVec() { free(p); }
    void resize(size_t count)
    {
        static_assert(std::is_trivially_copyable<T>::value && std::is_trivially_destructible<T>::value, "raw storage");
        free(p);
        p = count ? (T *)malloc(count * sizeof(T)) : nullptr;
        if (count && !p) abort();
        n = count;
    }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
    const T *data() const { return p; }

private:
    T *p = nullptr;
    size_t n = 0;
};

// a read-only view of consecutive elements (a whole RawVec or a part of one); the owner outlives it
template <typename T> struct Span {
    const T *p = nullptr;
    size_t n = 0;
    Span() = default;
    Span(const T *first, size_t count) : p(first), n(count) {}
    Span(const RawVec<T> &v) : p(v.data()), n(v.size()) {}
    size_t size() const { return n; }
    const T &operator[](size_t i) const { return p[i]; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
};

#endif
