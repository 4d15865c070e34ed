// FQEntry.h -- one FASTQ record as four views into its batch, plus the framing checks.
// Role of reference src/FQEntry.{h,cpp}; the checks and their messages (src/FQEntry.cpp:53-97)
// are part of the CLI's observable behaviour and are reproduced.
#ifndef SICKLE_FQENTRY_H
#define SICKLE_FQENTRY_H

#include <string_view>

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

#endif
