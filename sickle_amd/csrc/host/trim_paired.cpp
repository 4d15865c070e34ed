#include "trim_paired.h"

#include <getopt.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>

// reference src/trim_paired.cpp:16-36
static struct option paired_long_options[] = {
    {"qual-type", required_argument, 0, 't'},
    {"pe-file1", required_argument, 0, 'f'},
    {"pe-file2", required_argument, 0, 'r'},
    {"pe-interleaved", required_argument, 0, 'c'},
    {"output-pe1", required_argument, 0, 'o'},
    {"output-pe2", required_argument, 0, 'p'},
    {"output-single", required_argument, 0, 's'},
    {"output-interleaved", required_argument, 0, 'm'},
    {"qual-threshold", required_argument, 0, 'q'},
    {"length-threshold", required_argument, 0, 'l'},
    {"no-fiveprime", no_argument, 0, 'x'},
    {"truncate-n", no_argument, 0, 'n'},
    {"gzip-output", no_argument, 0, 'g'},
    {"quiet", no_argument, 0, 'z'},
    {"threads", no_argument, 0, 'a'},
    {"batch", no_argument, 0, 'b'},
    {"help", no_argument, NULL, CHAR_MIN - 2},
    {"version", no_argument, NULL, CHAR_MIN - 3},
    {NULL, 0, NULL, 0}};

// text of reference src/trim_paired.cpp:38-76
void Trim_Paired::usage(int status, char const *msg)
{
    fprintf(stderr, "\nIf you have separate files for forward and reverse reads:\n");
    fprintf(stderr, "Usage: %s pe [options] -f <paired-end forward fastq file> -r <paired-end reverse fastq file> -t <quality type> -o <trimmed PE forward file> -p <trimmed PE reverse file> -s <trimmed singles file>\n\n", PROGRAM_NAME);
    fprintf(stderr, "If you have one file with interleaved forward and reverse reads:\n");
    fprintf(stderr, "Usage: %s pe [options] -c <interleaved input file> -t <quality type> -m <interleaved trimmed paired-end output> -s <trimmed singles file>\n\n\
If you have one file with interleaved reads as input and you want ONLY one interleaved file as output:\n\
Usage: %s pe [options] -c <interleaved input file> -t <quality type> -m <interleaved trimmed output>\n\n", PROGRAM_NAME, PROGRAM_NAME);
    fprintf(stderr, "Options:\n\
Paired-end separated reads\n\
--------------------------\n\
-f, --pe-file1, Input paired-end forward fastq file (Input files must have same number of records)\n\
-r, --pe-file2, Input paired-end reverse fastq file\n\
-o, --output-pe1, Output trimmed forward fastq file\n\
-p, --output-pe2, Output trimmed reverse fastq file. Must use -s option.\n\n\
Paired-end interleaved reads\n\
----------------------------\n");
    fprintf(stderr, "-c, --pe-interleaved, Combined (interleaved) input paired-end fastq\n\
-m, --output-interleaved, Output combined (interleaved) paired-end fastq file. Must use -s option.\n\
--------------\n\
-t, --qual-type, Type of quality values (solexa (CASAVA < 1.3), illumina (CASAVA 1.3 to 1.7), sanger (which is CASAVA >= 1.8)) (required)\n");
    fprintf(stderr, "-s, --output-single, Output trimmed singles fastq file\n\
-q, --qual-threshold, Threshold for trimming based on average quality in a window. Default 20.\n\
-l, --length-threshold, Threshold to keep a read based on length after trimming. Default 20.\n\
-x, --no-fiveprime, Don't do five prime trimming.\n\
-n, --truncate-n, Truncate sequences at position of first N.\n\
-a, --threads, Number of threads to use. Default and minimum: Available cores - 1.\n\
-b, --batch, maximum MB of data to read from the input file at each cycle.\n\
\tThe greater the value, the greater the memory usage can be. The value, multiplied by 1024^2, must be \n\
\tbigger than the lenght of the longest read. Minimum 1. Default: 512.\n");

    fprintf(stderr, "-g, --gzip-output, Output gzipped files.\n--quiet, do not output trimming info\n\
--help, display this help and exit\n\
--version, output version information and exit\n\n");

    if (msg) fprintf(stderr, "%s\n\n", msg);
    exit(status);
}

Trim_Paired::Trim_Paired()
    : input2(nullptr), input_inter(nullptr), interleaved_s(0), outfn2(nullptr), outfnc(nullptr), sfn(nullptr),
      infn2(nullptr), infnc(nullptr), kept_p(0), discard_p(0), kept_s1(0), kept_s2(0), discard_s1(0), discard_s2(0)
{
    threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    batch_len = 1024 * 1024 * DEFAULT_BATCH_LEN;
}

static void print_version_and_exit()
{
    fprintf(stdout,
            "%s version %0.3f\nCopyright (c) 2011 The Regents "
            "of University of California, Davis Campus.\n"
            "%s is free software and comes with ABSOLUTELY NO WARRANTY.\n"
            "Distributed under the MIT License.\n\nWritten by %s\n",
            PROGRAM_NAME, VERSION, PROGRAM_NAME, AUTHORS);
    exit(EXIT_SUCCESS);
}

int Trim_Paired::parse_args(int argc, char *argv[])
{
    int optc;
    while (1) {
        int option_index = 0;
        // "M:" is accepted by the optstring but has no case: -M falls to usage (exit 1), as in the reference
        optc = getopt_long(argc, argv, "df:r:c:t:o:p:m:M:s:q:a:b:l:xng", paired_long_options, &option_index);
        if (optc == -1) break;
        switch (optc) {
        case 'f': infn = strdup(optarg); break;
        case 'r': infn2 = strdup(optarg); break;
        case 'c': infnc = strdup(optarg); break;
        case 't':
            if (!strcmp(optarg, "illumina")) qualtype = ILLUMINA;
            else if (!strcmp(optarg, "solexa")) qualtype = SOLEXA;
            else if (!strcmp(optarg, "sanger")) qualtype = SANGER;
            else {
                fprintf(stderr, "Error: Quality type '%s' is not a valid type.\n", optarg);
                return EXIT_FAILURE;
            }
            break;
        case 'o': outfn = strdup(optarg); break;
        case 'p': outfn2 = strdup(optarg); break;
        case 'm':
            outfnc = strdup(optarg);
            interleaved_s = 1;
            break;
        case 's': sfn = strdup(optarg); break;
        case 'q':
            qual_threshold = atoi(optarg);
            if (qual_threshold < 0) {
                fprintf(stderr, "Quality threshold must be >= 0\n");
                return EXIT_FAILURE;
            }
            break;
        case 'l':
            length_threshold = atoi(optarg);
            if (length_threshold < 0) {
                fprintf(stderr, "Length threshold must be >= 0\n");
                return EXIT_FAILURE;
            }
            break;
        case 'x': no_fiveprime = 1; break;
        case 'n': trunc_n = 1; break;
        case 'g': gzip_output = 1; break;
        case 'z': quiet = 1; break;
        case 'd': debug = 1; break;
        case 'a':
            if (!optarg) usage(EXIT_FAILURE, NULL); // long form has no argument: the reference crashes here
            threads = atoi(optarg);
            break;
        case 'b':
            if (!optarg) usage(EXIT_FAILURE, NULL);
            batch_len = 1024 * 1024 * (atoi(optarg));
            break;
        case CHAR_MIN - 2: usage(EXIT_SUCCESS, NULL); break;
        case CHAR_MIN - 3: print_version_and_exit(); break;
        case '?':
        default: usage(EXIT_FAILURE, NULL); break;
        }
    }

    if (qualtype == -1) {
        usage(EXIT_FAILURE, "****Error: Quality type is required.");
        return EXIT_FAILURE;
    }
    if (!infn && !infnc) {
        usage(EXIT_FAILURE, "****Error: Must have either -f OR -c argument.");
        return EXIT_FAILURE;
    }
    if (threads < 1) threads = 1;
    if (infnc) batch_len = recommended_batch_len(infnc, batch_len);
    else if (infn) batch_len = recommended_batch_len(infn, batch_len);
    return 0;
}

int Trim_Paired::recommended_batch_len(const char *path, int max_batch_len)
{
    return recommended_batch_len_for(path, (unsigned)max_batch_len / 2); // src/trim_paired.cpp:248
}

// reference src/trim_paired.cpp:626-731
int Trim_Paired::init_streams()
{
    const bool gz = gzip_output != 0;
    if (infnc) { /* interleaved input */
        if (infn || infn2 || outfn || outfn2) {
            usage(EXIT_FAILURE, "****Error: Cannot have -f, -r, -o, or -p options with -c.");
            return EXIT_FAILURE;
        }
        if (!outfnc) {
            // the reference opens a NULL file name here and dies; -m is required with -c
            usage(EXIT_FAILURE, "****Error: Using the -c option means you must have the -m option.");
            return EXIT_FAILURE;
        }
        input_inter = new GZReader(infnc, batch_len, true);
        if (!input_inter->is_open()) return EXIT_FAILURE;
        input = input_inter;
        if (!outfile_interleaved.open(outfnc, gz)) {
            fprintf(stderr, "****Error: Could not open interleaved output file '%s'.\n\n", outfnc);
            return EXIT_FAILURE;
        }
    } else { /* forward and reverse input files */
        if (infn && (!infn2 || !outfn || !outfn2 || !sfn)) {
            usage(EXIT_FAILURE, "****Error: Using the -f option means you must have the -r, -o, -p, and -s options.");
            return EXIT_FAILURE;
        }
        if (infn && (infnc || interleaved_s)) {
            usage(EXIT_FAILURE, "****Error: The -f option cannot be used in combination with -c, -m, or -M.");
            return EXIT_FAILURE;
        }
        staging_files = 2;
        input = new GZReader(infn, batch_len);
        if (!input->is_open()) return EXIT_FAILURE;
        input2 = new GZReader(infn2, batch_len);
        if (!input2->is_open()) return EXIT_FAILURE;
        if (!outfile.open(outfn, gz)) {
            fprintf(stderr, "****Error: Could not open output file '%s'.\n\n", outfn);
            return EXIT_FAILURE;
        }
        if (!outfile2.open(outfn2, gz)) {
            fprintf(stderr, "****Error: Could not open output file '%s'.\n\n", outfn2);
            return EXIT_FAILURE;
        }
    }
    if (sfn) {
        if (!outfile_single.open(sfn, gz)) {
            fprintf(stderr, "****Error: Could not open single output file '%s'.\n\n", sfn);
            return EXIT_FAILURE;
        }
    }
    return open_device();
}

void Trim_Paired::close_streams()
{
    outfile_single.close();
    outfile.close();
    outfile2.close();
    outfile_interleaved.close();
    if (sickle_leave_fast) return; // see sickle.h
    if (input_inter) {
        delete input_inter;
    } else {
        delete input;
        delete input2;
    }
    input = input2 = input_inter = nullptr;
    close_device();
}

// Pair classification and the three output streams of one ingest batch: reference
// src/trim_paired.cpp:515-624.  Pair k of the batch sits in queue k mod T and the queues are
// written one after the other (:388-403, :530-533), so -a T > 1 gives queue-major file order.
Trim_Paired::Assembled *Trim_Paired::output_paired(Work &w)
{
    const size_t pairs = w.reads.size() / 2;
    const size_t T = (size_t)threads;
    // queue q holds pairs k = q, q+T, ...; start[q] = output index of its first
    std::vector<size_t> start(T + 1, 0);
    for (size_t q = 0; q < T; ++q) start[q + 1] = start[q] + (q < pairs ? (pairs - q + T - 1) / T : 0);
    WorkerPool &pool = WorkerPool::instance();
    const size_t parts = std::min<size_t>((size_t)pool.size() * 2, pairs ? pairs : 1);
    struct Part {
        std::string fq1, fq2, singles;
        int kept_p = 0, kept_s1 = 0, kept_s2 = 0, discard_p = 0, discard_s1 = 0, discard_s2 = 0;
    };
    std::vector<Part> part_out(parts);
    const bool inter = input_inter != nullptr;
    pool.parallel_for(pairs, parts, [&](size_t lo, size_t hi, size_t part) {
        Part &o = part_out[part];
        size_t q = 0;
        while (start[q + 1] <= lo) ++q;
        size_t k = q + (lo - start[q]) * T;
        // reservation: a stretch keeps most of its bytes; sized from its first pair
        const FQEntry &f = w.reads[2 * k];
        const size_t approx = (hi - lo) * (f.name.size() + f.comment.size() + 2 * f.seq.size() + 8);
        o.fq1.reserve(inter ? 2 * approx : approx);
        if (!inter) o.fq2.reserve(approx);
        for (size_t j = lo; j < hi; ++j) {
            const FQEntry &read1 = w.reads[2 * k], &read2 = w.reads[2 * k + 1];
            const cutsites &cs1 = w.cuts[2 * k], &cs2 = w.cuts[2 * k + 1];
            const bool r1 = cs1.three_prime_cut >= 0; // src/trim_paired.cpp:500,502
            const bool r2 = cs2.three_prime_cut >= 0;
            if (r1 && r2) {
                append_record(o.fq1, read1, cs1);
                if (inter) append_record(o.fq1, read2, cs2);
                else append_record(o.fq2, read2, cs2);
                o.kept_p += 2;
            } else if (r1 || r2) {
                if (r1) {
                    append_record(o.singles, read1, cs1);
                    o.kept_s1++;
                    o.discard_s2++;
                } else {
                    append_record(o.singles, read2, cs2);
                    o.kept_s2++;
                    o.discard_s1++;
                }
            } else {
                o.discard_p += 2;
            }
            k += T;
            if (k >= pairs && j + 1 < hi) { ++q; while (start[q + 1] == start[q]) ++q; k = q; }
        }
    });
    int b_kept_p = 0, b_kept_s1 = 0, b_kept_s2 = 0, b_discard_p = 0, b_discard_s1 = 0, b_discard_s2 = 0;
    for (const Part &o : part_out) {
        b_kept_p += o.kept_p;
        b_kept_s1 += o.kept_s1;
        b_kept_s2 += o.kept_s2;
        b_discard_p += o.discard_p;
        b_discard_s1 += o.discard_s1;
        b_discard_s2 += o.discard_s2;
    }
    kept_p += b_kept_p;
    kept_s1 += b_kept_s1;
    kept_s2 += b_kept_s2;
    discard_p += b_discard_p;
    discard_s1 += b_discard_s1;
    discard_s2 += b_discard_s2;
    // src/trim_paired.cpp:593 computes `total` from the PER-BATCH locals that shadow the members,
    // so the summary's "Total input FastQ records" is the size of the last INGEST batch written
    // (a batch may arrive here in pieces at -a 1)
    if (w.first_of_batch) batch_total = 0;
    batch_total += b_kept_p + b_kept_s1 + b_kept_s2 + b_discard_p + b_discard_s1 + b_discard_s2;
    total = batch_total;

    Assembled *a = new Assembled();
    for (Part &o : part_out) {
        a->fq1.push_back(std::move(o.fq1));
        a->fq2.push_back(std::move(o.fq2));
        a->singles.push_back(std::move(o.singles));
    }
    w.frame.reset(); // the last piece of a batch releases its text and its record array
    return a;
}

void Trim_Paired::write_assembled(Assembled *a)
{
    // the output files are independent: write them side by side
    const bool inter = input_inter != nullptr;
    std::thread t2, t3;
    if (!inter)
        t2 = std::thread([&] { outfile2.write_parts(a->fq2); });
    if (sfn)
        t3 = std::thread([&] { outfile_single.write_parts(a->singles); });
    OutFile &first = inter ? outfile_interleaved : outfile;
    first.write_parts(a->fq1);
    if (t2.joinable()) t2.join();
    if (t3.joinable()) t3.join();
    delete a;
}

int Trim_Paired::trim_main()
{
    total = 0;
    kept_p = discard_p = kept_s1 = kept_s2 = discard_s1 = discard_s2 = 0;
    StageClock::mark("trim_main");
    int res = init_streams();
    if (res != 0) return res;
    StageClock::mark("streams + device open");

    Channel<Work *> parsed(2), scanned(2);
    StageClock clk_read, clk_frame, clk_pack, clk_wait, clk_out;
    // each input file is read and indexed ahead on its own thread
    Channel<Batch *> raw1(1), raw2(1);
    std::thread fetch1 = prefetch_batches(input, raw1), fetch2;
    if (!input_inter) fetch2 = prefetch_batches(input2, raw2);
    std::thread reader([&] {
        // the batch loop of reference src/trim_paired.cpp:280-453, minus the worker threads
        while (true) {
            StageClock::Scope rd(clk_read);
            Batch *batch = NULL;
            if (!raw1.pop(batch) || batch == NULL) break;
            Batch *batch2 = NULL;
            if (!input_inter) {
                if (!raw2.pop(batch2)) batch2 = NULL;
                if (batch2 == NULL) {
                    delete batch;
                    break;
                }
                if (batch2->n_lines() != batch->n_lines()) {
                    error("Batch2 and Batch1 have different lengths, exiting"); // :335-338
                    delete batch;
                    delete batch2;
                    break;
                }
            }
            rd.stop();
            StageClock::Scope fr(clk_frame);
            std::shared_ptr<Frame> frame = std::make_shared<Frame>();
            frame->batch = batch;
            frame->batch2 = batch2;
            // :350-404 -- pairs are framed until the batch ends or until the running sum of mate-1
            // lengths passes batch_len (the check sits BEFORE each pair, :352-358); what is left
            // of the batch after that is dropped
            const size_t lines_per_pair = input_inter ? 8 : 4;
            const size_t all_pairs = (size_t)batch->n_lines() / lines_per_pair;
            size_t pairs = 0;
            int chars_read_from_batch = 0;
            while (pairs < all_pairs) {
                if (chars_read_from_batch > batch_len) break;
                chars_read_from_batch += (int)batch->line(pairs * lines_per_pair + 1).length();
                ++pairs;
            }
            if (input_inter && pairs == all_pairs && (size_t)batch->n_lines() % 8 == 4) {
                // unreachable (interleaved batches hold whole pairs), kept for the message
                error("Reading interleaved pair: read1 loaded, but no read2 to load. Maybe it's not an interleaved file?");
                fatal_exit(EXIT_FAILURE);
            }
            frame->all.resize(2 * pairs);
            // record positions restart at 1 in every batch in PE (:309-310)
            if (input_inter) {
                frame_records(frame->all, *batch, 2 * pairs, [](size_t i) { return 4 * i; },
                              [](size_t i) { return (int)i + 1; });
            } else {
                // mate 1 first, then mate 2: a malformed mate 1 anywhere is reported before any
                // mate 2 problem only if it comes first in the reference's own order, which
                // alternates -- so check the pairs in that order afterwards
                RawVec<FQEntry> &rd = frame->all;
                WorkerPool &pool = WorkerPool::instance();
                const size_t parts = (size_t)pool.size() * 4;
                std::vector<size_t> first_bad(parts, (size_t)-1);
                pool.parallel_for(pairs, parts, [&](size_t lo, size_t hi, size_t part) {
                    for (size_t k = lo; k < hi; ++k) {
                        rd[2 * k] = FQEntry(*batch, 4 * k, (int)k + 1);
                        rd[2 * k + 1] = FQEntry(*batch2, 4 * k, (int)k + 1);
                        if ((!rd[2 * k].well_formed() || !rd[2 * k + 1].well_formed()) && first_bad[part] == (size_t)-1)
                            first_bad[part] = k;
                    }
                });
                for (size_t part = 0; part < parts; ++part)
                    if (first_bad[part] != (size_t)-1) {
                        rd[2 * first_bad[part]].validate(); // prints and exits if mate 1 is the bad one
                        rd[2 * first_bad[part] + 1].validate();
                        break;
                    }
            }
            if (chars_read_from_batch == 0) break; // :407-409 (the frame goes with its text)
            fr.stop();
            StageClock::mark("batch framed");
            // whole, or at -a 1 in pieces of whole pairs (trim.h: piece_reads)
            const size_t step = std::max<size_t>(1, piece_reads() / 2);
            for (size_t lo = 0; lo < pairs; lo += std::min(step, pairs - lo)) {
                const size_t hi = lo + std::min(step, pairs - lo);
                Work *w = new Work();
                w->frame = frame;
                w->reads = Span<FQEntry>(frame->all.data() + 2 * lo, 2 * (hi - lo));
                w->first_of_batch = lo == 0;
                parsed.push(w);
            }
        }
        parsed.close();
        // the run may end before the files do (different batch lengths, :335-338): drain the
        // prefetchers so that they can finish
        Batch *rest;
        while (raw1.pop(rest)) delete rest;
        if (!input_inter)
            while (raw2.pop(rest)) delete rest;
    });
    // assembly of batch i+1 overlaps the file writes of batch i
    Channel<Assembled *> assembled(1);
    StageClock clk_write;
    std::thread writer([&] {
        Work *w;
        while (scanned.pop(w)) {
            StageClock::Scope o(clk_out);
            Assembled *a = output_paired(*w);
            delete w;
            o.stop();
            assembled.push(a);
        }
        assembled.close();
    });
    std::thread flusher([&] {
        Assembled *a;
        while (assembled.pop(a)) {
            StageClock::Scope o(clk_write);
            write_assembled(a);
            StageClock::mark("batch written");
        }
    });

    const int nslots = n_slots();
    std::vector<Work *> inflight((size_t)nslots, nullptr);
    auto finish = [&](int slot) {
        Work *w = inflight[(size_t)slot];
        if (!w) return;
        const cutsites *cs;
        {
            StageClock::Scope wt(clk_wait);
            cs = wait_scan(slot, w->reads);
        }
        w->cuts.assign(cs, cs + w->reads.size());
        inflight[(size_t)slot] = nullptr;
        scanned.push(w);
    };
    int i = 0;
    Work *w;
    while (parsed.pop(w)) {
        const int slot = i % nslots;
        finish(slot);
        {
            StageClock::Scope pk(clk_pack);
            submit_scan(slot, w->reads);
        }
        StageClock::mark("batch submitted");
        inflight[(size_t)slot] = w;
        ++i;
    }
    for (int k = 0; k < nslots; ++k) finish((i + k) % nslots);
    require_device(); // even an empty input does not succeed without the GPU
    scanned.close();
    reader.join();
    fetch1.join();
    if (fetch2.joinable()) fetch2.join();
    writer.join();
    flusher.join();
    StageClock::report({{"read+index", &clk_read}, {"frame", &clk_frame}, {"pack+submit", &clk_pack},
                        {"device wait", &clk_wait}, {"classify+assemble", &clk_out}, {"write", &clk_write}});

    if (!quiet) { // reference src/trim_paired.cpp:464-476
        if (infn && infn2) fprintf(stdout, "\nPE forward file: %s\nPE reverse file: %s\n", infn, infn2);
        if (infnc) fprintf(stdout, "\nPE interleaved file: %s\n", infnc);
        fprintf(stdout, "\nTotal input FastQ records: %d (%d pairs)\n", total, (total / 2));
        fprintf(stdout, "\nFastQ paired records kept: %d (%d pairs)\n", kept_p, (kept_p / 2));
        if (input_inter) fprintf(stdout, "FastQ single records kept: %d\n", (kept_s1 + kept_s2));
        else fprintf(stdout, "FastQ single records kept: %d (from PE1: %d, from PE2: %d)\n", (kept_s1 + kept_s2), kept_s1, kept_s2);
        fprintf(stdout, "FastQ paired records discarded: %d (%d pairs)\n", discard_p, (discard_p / 2));
        if (input_inter) fprintf(stdout, "FastQ single records discarded: %d\n\n", (discard_s1 + discard_s2));
        else fprintf(stdout, "FastQ single records discarded: %d (from PE1: %d, from PE2: %d)\n\n", (discard_s1 + discard_s2), discard_s1, discard_s2);
    }
    close_streams();
    StageClock::mark("closed");
    return EXIT_SUCCESS;
}
