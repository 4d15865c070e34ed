#include "trim_single.h"

#include <getopt.h>

#include <climits>

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <thread>

// reference src/trim_single.cpp:20-35 (--threads/--batch are declared without an argument
// there too, so only the short forms -a N / -b N carry a value)
static struct option single_long_options[] = {
    {"fastq-file", required_argument, 0, 'f'},
    {"output-file", required_argument, 0, 'o'},
    {"qual-type", required_argument, 0, 't'},
    {"qual-threshold", required_argument, 0, 'q'},
    {"length-threshold", required_argument, 0, 'l'},
    {"no-fiveprime", no_argument, 0, 'x'},
    {"discard-n", no_argument, 0, 'n'},
    {"gzip-output", no_argument, 0, 'g'},
    {"quiet", no_argument, 0, 'z'},
    {"threads", no_argument, 0, 'a'},
    {"batch", no_argument, 0, 'b'},
    {"help", no_argument, NULL, CHAR_MIN - 2},
    {"version", no_argument, NULL, CHAR_MIN - 3},
    {NULL, 0, NULL, 0}};

// text of reference src/trim_single.cpp:37-61
void Trim_Single::usage(int status, char const *msg)
{
    fprintf(stderr, "\nUsage: %s se [options] -f <fastq sequence file> -t <quality type> -o <trimmed fastq file>\n\
\n\
Options:\n\
-f, --fastq-file, Input fastq file (required)\n\
-t, --qual-type, Type of quality values (solexa (CASAVA < 1.3), illumina (CASAVA 1.3 to 1.7), sanger (which is CASAVA >= 1.8)) (required)\n\
-o, --output-file, Output trimmed fastq file (required)\n", PROGRAM_NAME);

    fprintf(stderr, "-q, --qual-threshold, Threshold for trimming based on average quality in a window. Default 20.\n\
-l, --length-threshold, Threshold to keep a read based on length after trimming. Default 20.\n\
-x, --no-fiveprime, Don't do five prime trimming.\n\
-n, --trunc-n, Truncate sequences at position of first N.\n\
-g, --gzip-output, Output gzipped files.\n\
-a, --threads, Number of threads to use. Default and minimum: Available cores - 1.\n\
-b, --batch, maximum MB of data to read from the input file at each cycle.\n\
\tThe greater the value, the greater the memory usage can be. The value, multiplied by 1024^2, must be \n\
\tbigger than the lenght of the longest read. Minimum 1. Default: 512.\n\
--quiet, Don't print out any trimming information\n\
--help, display this help and exit\n\
--version, output version information and exit\n\n");

    if (msg) fprintf(stderr, "%s\n\n", msg);
    exit(status);
}

Trim_Single::Trim_Single()
{
    threads = (int)std::thread::hardware_concurrency(); // DEFAULT_THREADS, reference src/sickle.h:23-25
    if (threads < 1) threads = 1;
    batch_len = 1024 * 1024 * DEFAULT_BATCH_LEN;
}

static void print_version_and_exit()
{
    // case_GETOPT_VERSION_CHAR, reference src/sickle.h:51-58
    fprintf(stdout,
            "%s version %0.3f\nCopyright (c) 2011 The Regents "
            "of University of California, Davis Campus.\n"
            "%s is free software and comes with ABSOLUTELY NO WARRANTY.\n"
            "Distributed under the MIT License.\n\nWritten by %s\n",
            PROGRAM_NAME, VERSION, PROGRAM_NAME, AUTHORS);
    exit(EXIT_SUCCESS);
}

int Trim_Single::parse_args(int argc, char *argv[])
{
    int optc;
    while (1) {
        int option_index = 0;
        optc = getopt_long(argc, argv, "df:t:o:q:a:b:l:zxng", single_long_options, &option_index);
        if (optc == -1) break;
        switch (optc) {
        case 'f':
            infn = strdup(optarg);
            break;
        case 't':
            if (!strcmp(optarg, "illumina")) qualtype = ILLUMINA;
            else if (!strcmp(optarg, "solexa")) qualtype = SOLEXA;
            else if (!strcmp(optarg, "sanger")) qualtype = SANGER;
            else {
                fprintf(stderr, "Error: Quality type '%s' is not a valid type.\n", optarg);
                return EXIT_FAILURE;
            }
            break;
        case 'o':
            outfn = strdup(optarg);
            break;
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
            // the long form carries no argument (optarg NULL): the reference crashes in atoi there
            if (!optarg) usage(EXIT_FAILURE, NULL);
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

    if (qualtype == -1 || !infn || !outfn)
        usage(EXIT_FAILURE, "****Error: Must have quality type, input file, and output file.");

    if (!strcmp(infn, outfn)) {
        fprintf(stderr, "****Error: Input file is same as output file.\n\n");
        return EXIT_FAILURE;
    }
    if (threads < 1) threads = 1; // the reference would spawn no worker and write nothing
    batch_len = recommended_batch_len(infn, batch_len);
    return 0;
}

int Trim_Single::recommended_batch_len(const char *path, int max_batch_len)
{
    return recommended_batch_len_for(path, (unsigned)max_batch_len); // src/trim_single.cpp:194-211
}

int Trim_Single::init_streams()
{
    input = new GZReader(infn, batch_len, false);
    if (!input->is_open()) return EXIT_FAILURE; // message printed by the reader
    if (!outfile.open(outfn, gzip_output != 0)) {
        fprintf(stderr, "****Error: Could not open output file '%s'.\n\n", outfn);
        return EXIT_FAILURE;
    }
    return open_device();
}

void Trim_Single::close_streams()
{
    outfile.close();
    if (sickle_leave_fast) return; // see sickle.h
    delete input;
    input = nullptr;
    close_device();
}

// One ingest batch -> output text.  The reference deals read k of a batch into queue (k+1) mod T
// and writes the queues one after the other (src/trim_single.cpp:263-298, :382-405), so with
// -a T > 1 the records of a batch come out queue-major.  Reproduced, since it is the file order.
std::vector<std::string> *Trim_Single::output_single(Work &w)
{
    const size_t n = w.reads.size();
    const size_t T = (size_t)threads;
    // queue q holds reads k = (q + T - 1) % T, +T, +2T, ...; start[q] = output index of its first
    std::vector<size_t> start(T + 1, 0);
    for (size_t q = 0; q < T; ++q) {
        const size_t k0 = (q + T - 1) % T;
        start[q + 1] = start[q] + (k0 < n ? (n - k0 + T - 1) / T : 0);
    }
    // every host thread assembles a contiguous stretch of the OUTPUT order into its own buffer
    WorkerPool &pool = WorkerPool::instance();
    const size_t parts = std::min<size_t>((size_t)pool.size() * 2, n ? n : 1);
    std::vector<std::string> text(parts);
    std::vector<int> part_kept(parts, 0), part_discard(parts, 0);
    pool.parallel_for(n, parts, [&](size_t lo, size_t hi, size_t part) {
        std::string &out = text[part];
        size_t bytes = 0;
        size_t q = 0;
        while (start[q + 1] <= lo) ++q;
        size_t k = (q + T - 1) % T + (lo - start[q]) * T;
        for (size_t j = lo; j < hi; ++j) { // size pass: reserve once
            const cutsites &cs = w.cuts[k];
            if (cs.three_prime_cut >= 0)
                bytes += w.reads[k].name.size() + w.reads[k].comment.size() + 4 +
                         2 * (size_t)(cs.three_prime_cut - cs.five_prime_cut);
            k += T;
            if (k >= n && j + 1 < hi) { ++q; while (start[q + 1] == start[q]) ++q; k = (q + T - 1) % T; }
        }
        out.reserve(bytes);
        q = 0;
        while (start[q + 1] <= lo) ++q;
        k = (q + T - 1) % T + (lo - start[q]) * T;
        for (size_t j = lo; j < hi; ++j) {
            const cutsites &cs = w.cuts[k];
            if (!(cs.three_prime_cut >= 0)) { // src/trim_single.cpp:368
                part_discard[part]++;
            } else {
                append_record(out, w.reads[k], cs);
                part_kept[part]++;
            }
            k += T;
            if (k >= n && j + 1 < hi) { ++q; while (start[q + 1] == start[q]) ++q; k = (q + T - 1) % T; }
        }
    });
    for (size_t part = 0; part < parts; ++part) {
        kept += part_kept[part];
        discard += part_discard[part];
    }
    total = kept + discard;
    w.frame.reset();
    return new std::vector<std::string>(std::move(text));
}

int Trim_Single::trim_main()
{
    kept = 0;
    discard = 0;
    total = 0;
    int res = init_streams();
    if (res != 0) return res;

    // ingest thread -> (this thread: device) -> output thread; two batches in flight on the
    // device so that the H2D copy of one overlaps the scan of the other
    Channel<Work *> parsed(2), scanned(2);
    Channel<Batch *> raw(1);
    std::thread fetcher = prefetch_batches(input, raw);
    std::thread reader([&] {
        int last_read_position = 0; // counts across batches in SE (src/trim_single.cpp:238)
        Batch *batch;
        while (raw.pop(batch) && batch) {
            std::shared_ptr<Frame> frame = std::make_shared<Frame>();
            frame->batch = batch;
            const size_t n = (size_t)batch->n_lines() / 4;
            frame->all.resize(n);
            const int base = last_read_position;
            frame_records(frame->all, *batch, n, [](size_t i) { return 4 * i; },
                          [base](size_t i) { return base + (int)i + 1; });
            last_read_position += (int)n;
            const size_t step = std::max<size_t>(1, piece_reads());
            for (size_t lo = 0; lo < n; lo += std::min(step, n - lo)) {
                const size_t hi = lo + std::min(step, n - lo);
                Work *w = new Work();
                w->frame = frame;
                w->reads = Span<FQEntry>(frame->all.data() + lo, hi - lo);
                parsed.push(w);
            }
        }
        parsed.close();
    });
    // assembly of batch i+1 overlaps the file write of batch i
    Channel<std::vector<std::string> *> assembled(1);
    std::thread writer([&] {
        Work *w;
        while (scanned.pop(w)) {
            assembled.push(output_single(*w));
            delete w;
        }
        assembled.close();
    });
    std::thread flusher([&] {
        std::vector<std::string> *text;
        while (assembled.pop(text)) {
            outfile.write_parts(*text);
            delete text;
        }
    });

    const int nslots = n_slots();
    std::vector<Work *> inflight((size_t)nslots, nullptr);
    auto finish = [&](int slot) {
        Work *w = inflight[(size_t)slot];
        if (!w) return;
        const cutsites *cs = wait_scan(slot, w->reads);
        w->cuts.assign(cs, cs + w->reads.size());
        inflight[(size_t)slot] = nullptr;
        scanned.push(w);
    };
    int i = 0;
    Work *w;
    while (parsed.pop(w)) {
        const int slot = i % nslots;
        finish(slot);
        submit_scan(slot, w->reads);
        inflight[(size_t)slot] = w;
        ++i;
    }
    for (int k = 0; k < nslots; ++k) finish((i + k) % nslots);
    require_device(); // even an empty input does not succeed without the GPU
    scanned.close();
    reader.join();
    fetcher.join();
    writer.join();
    flusher.join();

    if (!quiet)
        fprintf(stdout, "\nSE input file: %s\n\nTotal FastQ records: %d\nFastQ records kept: %d\nFastQ records discarded: %d\n\n",
                infn, total, kept, discard);
    close_streams();
    return EXIT_SUCCESS;
}
