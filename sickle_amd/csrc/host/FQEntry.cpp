#include "FQEntry.h"

#include <cstdlib>
#include <string>

#include "sickle.h"

using std::string;

FQEntry::FQEntry(int previous, Batch *reader)
{
    position = previous + 1;
    name = reader->next_line();
    seq = reader->next_line();
    comment = reader->next_line();
    qual = reader->next_line();
    validate();
}

FQEntry::FQEntry(const Batch &batch, size_t first_line, int position_) : position(position_)
{
    name = batch.line(first_line);
    seq = batch.line(first_line + 1);
    comment = batch.line(first_line + 2);
    qual = batch.line(first_line + 3);
}

bool FQEntry::well_formed() const
{
    return name.length() > 1 && name[0] == '@' && seq.length() >= 1 && qual.length() >= 1 &&
           qual.length() == seq.length();
}

// reference src/FQEntry.cpp:53-97
void FQEntry::validate() const
{
    if (name.length() <= 1) {
        error(string("In ") + string(name) + string("(line ") + std::to_string((position * 4) - 4) + string(")"));
        error("Sequence ID is to short.");
        error(string("ID:") + string(name));
        error(string("Sequence: ") + string(seq));
        error(string("Comment: ") + string(comment));
        error(string("Qualities: ") + string(qual));
        fatal_exit(EXIT_FAILURE);
    }
    if (name[0] != '@') {
        error(string("In ") + string(name) + string("(line ") + std::to_string((position * 4) - 4) + string(")"));
        error("Invalid char at the beggining of ID.");
        error(string("Sequence: ") + string(seq));
        error(string("Comment: ") + string(comment));
        error(string("Qualities: ") + string(qual));
        fatal_exit(EXIT_FAILURE);
    }
    if (seq.length() < 1) {
        error("Sequence line is empty");
        fatal_exit(EXIT_FAILURE);
    }
    if (qual.length() < 1) {
        error("Quality line is empty.");
        fatal_exit(EXIT_FAILURE);
    }
    if (qual.length() != seq.length()) {
        error("Sequence and quality lines have different lengths:");
        error(string(seq));
        error(string(qual));
        fatal_exit(EXIT_FAILURE);
    }
}
