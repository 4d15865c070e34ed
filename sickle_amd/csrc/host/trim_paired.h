// trim_paired.h -- `sickle pe`.  Role of reference src/trim_paired.{h,cpp}: same class name,
// entry points, options, messages, counters and exit codes.
#ifndef SICKLE_TRIM_PAIRED_H
#define SICKLE_TRIM_PAIRED_H

#include "trim.h"

class Trim_Paired : public Abstract_Trimmer {
public:
    Trim_Paired();
    int parse_args(int argc, char *argv[]) override;
    int trim_main() override;
    void usage(int status, char const *msg) override;
    int recommended_batch_len(const char *path, int max_batch_len);

protected:
    // one ingest batch, framed: owns the text (views of the mapped input) and the record array
    struct Frame {
        Batch *batch = nullptr, *batch2 = nullptr;
        RawVec<FQEntry> all; // mate 1 of pair k at 2k, mate 2 at 2k+1 (the reference's scan order)
        ~Frame()
        {
            delete batch;
            delete batch2;
        }
    };
    // what travels through the device and output stages: a whole frame, or at -a 1 a piece of one
    struct Work {
        std::shared_ptr<Frame> frame;
        Span<FQEntry> reads;
        std::vector<cutsites> cuts;
        bool first_of_batch = true;
    };
    int batch_total = 0; // records of the ingest batch being written (for the summary's "Total", see output_paired)
    int init_streams();
    void close_streams();
    // the text of one batch for the three outputs, in pieces (one per host thread), in order
    struct Assembled {
        std::vector<std::string> fq1, fq2, singles;
    };
    // classifies the pairs of a batch and builds their output text; updates the counters
    Assembled *output_paired(Work &w);
    void write_assembled(Assembled *a); // the three files side by side; deletes a

    GZReader *input2;
    GZReader *input_inter;
    OutFile outfile, outfile2, outfile_interleaved, outfile_single;
    int interleaved_s;
    char *outfn2; /* reverse file out name */
    char *outfnc; /* interleaved file out name */
    char *sfn;    /* singles file out name */
    char *infn2;  /* reverse input filename */
    char *infnc;  /* interleaved input filename */
    int kept_p, discard_p, kept_s1, kept_s2, discard_s1, discard_s2;
};

#endif
