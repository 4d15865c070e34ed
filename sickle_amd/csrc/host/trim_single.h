// trim_single.h -- `sickle se`.  Role of reference src/trim_single.{h,cpp}: same class name,
// same entry points (parse_args / trim_main / usage), same options, messages and exit codes.
#ifndef SICKLE_TRIM_SINGLE_H
#define SICKLE_TRIM_SINGLE_H

#include "trim.h"

class Trim_Single : public Abstract_Trimmer {
public:
    Trim_Single();
    int parse_args(int argc, char *argv[]) override;
    int recommended_batch_len(const char *path, int max_len);
    int trim_main() override;
    void usage(int status, char const *msg) override;
    int init_streams();
    void close_streams();

private:
    struct Frame { // one ingest batch, framed: owns the text and the record array
        Batch *batch = nullptr;
        RawVec<FQEntry> all;
        ~Frame() { delete batch; }
    };
    struct Work { // a whole frame, or at -a 1 a piece of one (trim.h: piece_reads)
        std::shared_ptr<Frame> frame;
        Span<FQEntry> reads;
        std::vector<cutsites> cuts;
    };
    // builds the output text of one batch, in pieces, in file order; updates the counters
    std::vector<std::string> *output_single(Work &w);
    OutFile outfile;
};

#endif
