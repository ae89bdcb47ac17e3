// test_host_mirror.cpp -- the C++ host mirror used the way the reference's
// callers use the reference classes:
//   * `brutefir` like foo_dsp_bfir.cpp:279-345 (construct, set_coeff, run per block)
//   * `fftw_convolver` + coeff::preprocess_coeff driven through the exact call
//     sequence of brutefir::run (brutefir.cpp:252-334)
// Both are checked against a long-double direct-form convolution computed here.
// Build: g++ -std=c++17 tests/cpp/test_host_mirror.cpp -Lfoo-dsp-bfir_amd/lib -lbfir_hip
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../foo-dsp-bfir_amd/host/brutefir_hip.hpp"
#include "../../foo-dsp-bfir_amd/host/coeff_hip.hpp"
#include "../../foo-dsp-bfir_amd/host/preprocessor_hip.hpp"
#include "../../foo-dsp-bfir_amd/host/equalizer_hip.hpp"

static int g_fail = 0;
#define CHECK(cond, ...)                                        \
    do {                                                        \
        if (!(cond)) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); g_fail++; } \
    } while (0)

static void log_line(const char *m) { printf("[log] %s\n", m); }

template <typename T> static double run_case(int L, int B, int C, int taps, int nb)
{
    const int realsize = sizeof(T);
    const int fmt = realsize == 4 ? BF_SAMPLE_FORMAT_FLOAT_LE : BF_SAMPLE_FORMAT_FLOAT64_LE;
    std::mt19937_64 rng(L * 31 + B * 7 + C);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    std::vector<std::vector<T>> h(C, std::vector<T>(taps));
    for (int c = 0; c < C; c++) {
        double sum = 0;
        for (int n = 0; n < taps; n++) { double v = u(rng) * std::exp(-6.0 * n / taps); h[c][n] = (T)v; sum += std::fabs(v); }
        for (int n = 0; n < taps; n++) h[c][n] = (T)(h[c][n] / sum);
    }
    std::vector<T> x((size_t)nb * L * C), y(x.size()), y2(x.size());
    for (auto &v : x) v = (T)u(rng);

    // ---- the engine class, one run() per block like dsp_bfir::on_chunk ----
    brutefir filter(L, B, realsize, C, fmt, fmt, 44100, false);
    filter.set_log(log_line);
    CHECK(filter.create_error() == 0, "create: %s", bfir_strerror(filter.create_error()));
    CHECK(!filter.is_initialized(), "initialised before set_coeff");
    CHECK(filter.run(x.data(), y.data()) != 0, "run without coefficients must fail");
    std::vector<void *> hp(C);
    for (int c = 0; c < C; c++) hp[c] = h[c].data();
    CHECK(filter.set_coeff(hp.data(), C, taps, B, 1.0) == 0, "set_coeff");
    CHECK(filter.is_initialized(), "not initialised after set_coeff");
    for (int t = 0; t < nb; t++)
        CHECK(filter.run(&x[(size_t)t * L * C], &y[(size_t)t * L * C]) == 0, "run block %d", t);
    filter.check_overflows();

    // ---- reference: direct-form convolution in long double ----
    double maxref = 0, maxerr = 0;
    std::vector<double> ref(x.size());
    const int used = std::min(taps, B * L);
    for (int c = 0; c < C; c++)
        for (int n = 0; n < nb * L; n++) {
            long double acc = 0;
            for (int k = 0; k <= std::min(n, used - 1); k++) acc += (long double)h[c][k] * (long double)x[(size_t)(n - k) * C + c];
            ref[(size_t)n * C + c] = (double)acc;
            maxref = std::max(maxref, std::fabs((double)acc));
        }
    for (size_t i = 0; i < x.size(); i++) maxerr = std::max(maxerr, std::fabs((double)y[i] - ref[i]));
    const double tol = realsize == 4 ? 1e-5 : 1e-12;
    CHECK(maxerr / maxref <= tol, "brutefir vs direct: %.3g", maxerr / maxref);

    // ---- the convolver class through brutefir::run's own call sequence ----
    fftw_convolver conv(L, realsize, nullptr);
    const int cb = conv.convolver_cbufsize();
    CHECK(cb == 2 * L * realsize, "cbufsize %d", cb);
    std::vector<void **> coeffs(C);
    for (int c = 0; c < C; c++) coeffs[c] = coeff::preprocess_coeff(&conv, h[c].data(), L, B, taps, realsize, 1.0);
    auto newbuf = [&]() { void *p = bfir_aligned_malloc(cb, 16); memset(p, 0, cb); return p; };
    std::vector<std::vector<void *>> fdl(C, std::vector<void *>(B));
    std::vector<void *> ocbuf(C), ifreq(C), ofreq(C);
    std::vector<std::vector<void *>> tbuf(C, std::vector<void *>(2));
    for (int c = 0; c < C; c++) {
        for (int b = 0; b < B; b++) fdl[c][b] = newbuf();
        ocbuf[c] = (B > 1) ? newbuf() : fdl[c][0];
        tbuf[c][0] = newbuf(); tbuf[c][1] = newbuf(); ifreq[c] = newbuf(); ofreq[c] = newbuf();
    }
    buffer_format_t bfs[BF_MAXCHANNELS];
    bfoverflow_t ofl[BF_MAXCHANNELS];
    for (int c = 0; c < C; c++) {   // setup_input / setup_output, brutefir.cpp:512-582
        bfs[c].sf.isfloat = true; bfs[c].sf.swap = false; bfs[c].sf.bytes = bfs[c].sf.sbytes = realsize;
        bfs[c].sf.scale = 1.0; bfs[c].sf.format = fmt;
        bfs[c].byte_offset = c * realsize; bfs[c].sample_spacing = C;
        ofl[c].n_overflows = 0; ofl[c].intlargest = 0; ofl[c].largest = 0; ofl[c].max = 1.0;
    }
    int curbuf = 0; unsigned blockcounter = 0; std::vector<int> proc(C, 0);
    for (int t = 0; t < nb; t++) {
        void *inbuf = &x[(size_t)t * L * C], *outbuf = &y2[(size_t)t * L * C];
        for (int n = 0; n < C; n++) {
            conv.convolver_raw2cbuf(inbuf, tbuf[n][curbuf], tbuf[n][!curbuf], &bfs[n], NULL, NULL);
            conv.convolver_time2freq(tbuf[n][curbuf], ifreq[n]);
            if (proc[n] < B) proc[n]++;
            int curblock = (int)(blockcounter % (unsigned)B);
            conv.convolver_mixnscale(&ifreq[n], fdl[n][curblock], &bfs[n].sf.scale, 1, CONVOLVER_MIXMODE_INPUT);
            if (B == 1) {
                conv.convolver_convolve_inplace(fdl[n][0], coeffs[n][0]);
            } else {
                conv.convolver_convolve(fdl[n][curblock], coeffs[n][0], ocbuf[n]);
                for (int i = 1; i < B && i < proc[n]; i++)
                    conv.convolver_convolve_add(fdl[n][(blockcounter - i) % (unsigned)B], coeffs[n][i], ocbuf[n]);
            }
            conv.convolver_mixnscale(&ocbuf[n], ofreq[n], &bfs[n].sf.scale, 1, CONVOLVER_MIXMODE_OUTPUT);
            // the reference reuses ocbuf[0] as the time-domain scratch (brutefir.cpp:311); ifreq[n] is free here
            conv.convolver_freq2time(ofreq[n], ifreq[n]);
            conv.convolver_cbuf2raw(ifreq[n], outbuf, &bfs[n], false, NULL, &ofl[n]);
            CHECK(conv.last_status() == 0, "stage call failed: %s", bfir_strerror(conv.last_status()));
        }
        curbuf = !curbuf; blockcounter++;
    }
    double e2 = 0;
    for (size_t i = 0; i < x.size(); i++) e2 = std::max(e2, std::fabs((double)y2[i] - ref[i]));
    CHECK(e2 / maxref <= tol, "facade sequence vs direct: %.3g", e2 / maxref);
    bfir_overflow eo;
    bfir_engine_get_overflow(filter.handle(), 0, &eo);
    CHECK(std::fabs(eo.largest - ofl[0].largest) <= tol * std::max(1e-30, ofl[0].largest), "peak %g vs %g", eo.largest, ofl[0].largest);
    for (int c = 0; c < C; c++) {
        for (int b = 0; b < B; b++) { bfir_aligned_free(coeffs[c][b]); bfir_aligned_free(fdl[c][b]); }
        bfir_aligned_free(coeffs[c]);
        if (B > 1) bfir_aligned_free(ocbuf[c]);
        bfir_aligned_free(tbuf[c][0]); bfir_aligned_free(tbuf[c][1]); bfir_aligned_free(ifreq[c]); bfir_aligned_free(ofreq[c]);
    }
    printf("L=%d B=%d C=%d realsize=%d: engine %.3g, facade %.3g (rel to max)\n", L, B, C, realsize,
           maxerr / maxref, e2 / maxref);
    return maxerr / maxref;
}

int main()
{
    if (bfir_device_count() < 1) { printf("no HIP device\n"); return 2; }
    run_case<float>(1024, 4, 2, 3900, 9);
    run_case<double>(1024, 4, 2, 3900, 9);
    run_case<float>(256, 1, 3, 200, 5);      // single partition: convolve_inplace branch
    run_case<double>(4096, 2, 8, 8192, 4);
    // error conventions
    {
        brutefir f(1024, 2, 4, 2, BF_SAMPLE_FORMAT_FLOAT_LE, BF_SAMPLE_FORMAT_FLOAT_LE, 44100, false);
        std::vector<float> a(100, 0.1f), b(100, 0.1f);
        b[10] = NAN;
        void *p[2] = {a.data(), b.data()};
        CHECK(f.set_coeff(p, 2, 100, 2, 1.0) == -2, "NaN tap must give -2");
        CHECK(!f.is_initialized(), "must stay uninitialised");
        brutefir bad(1000, 2, 4, 2, BF_SAMPLE_FORMAT_FLOAT_LE, BF_SAMPLE_FORMAT_FLOAT_LE, 44100, false);
        CHECK(bad.create_error() != 0 && !bad.is_initialized(), "length 1000 must be rejected");
        bool threw = false;
        try { fftw_convolver c(1000, 4, nullptr); } catch (...) { threw = true; }
        CHECK(threw, "convolver ctor must throw on a bad length");
    }
    // offline drivers (preprocessor.cpp): a pure-gain impulse makes the expected answers exact
    {
        const int L = 256, C = 2, n = 600;
        std::vector<float> ir((size_t)n * C, 0.f), noise((size_t)768 * C);
        ir[0] = 4.0f; ir[1] = 2.0f;                       // channel gains 4 and 2 at tap 0
        std::mt19937 rng(7); std::uniform_real_distribution<float> u(-1.f, 1.f);
        float peak = 0;
        for (size_t i = 0; i < noise.size(); i++) { noise[i] = u(rng); if (i % C == 0) peak = std::max(peak, std::fabs(noise[i])); }
        double att = 1;
        CHECK(preprocessor::calculate_attenuation(ir.data(), C, n, L, 4, noise.data(), &att), "calculate_attenuation");
        CHECK(std::fabs(att + 20.0 * std::log10(4.0 * peak)) < 1e-4, "attenuation %g vs %g", att, -20.0 * std::log10(4.0 * peak));
        std::vector<float> a((size_t)300 * C, 0.f), b((size_t)500 * C, 0.f);
        a[0] = 0.5f; a[1] = 0.25f; b[2 * C] = 1.0f; b[2 * C + 1] = 1.0f;   // gain, then a 2-frame delay
        std::vector<preprocessor::impulse_array> imps = {{a.data(), C, 300, 3.0}, {b.data(), C, 500, 1.0}};
        std::vector<uint8_t> out; int oc = 0, of = 0;
        CHECK(preprocessor::convolve_impulses(imps, L, 4, &out, &oc, &of), "convolve_impulses");
        CHECK(oc == C && of == 500, "shape %d x %d", of, oc);
        const float *o = (const float *)out.data();
        // pass 1: dirac * a = a; coefficients become 3*a; pass 2: b * (3a) = 1.5 / 0.75 at frame 2
        CHECK(std::fabs(o[2 * C] - 1.5f) < 1e-5 && std::fabs(o[2 * C + 1] - 0.75f) < 1e-5, "cascade %g %g", o[2 * C], o[2 * C + 1]);
        double rest = 0;
        for (int i = 0; i < 500 * C; i++) if (i != 2 * C && i != 2 * C + 1) rest = std::max(rest, (double)std::fabs(o[i]));
        CHECK(rest < 1e-5, "cascade residue %g", rest);
    }
    // equalizer (equalizer.cpp): a flat 0 dB / zero-phase EQ renders to a unit impulse; cache file round trip
    {
        const char *dir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
        equalizer eq(1024, 8, 8, 2, 48000);
        double f[1] = {1000.0}, m[1] = {0.0}, ph[1] = {0.0};
        std::string path = eq.generate(1, f, m, ph, dir);
        CHECK(!path.empty(), "equalizer render failed");
        std::vector<uint8_t> frames; int ch = 0, nf = 0, rs = 0, rate = 0;
        CHECK(wav_io::load_float(path, &frames, &ch, &nf, &rs, &rate), "cache WAV unreadable: %s", path.c_str());
        CHECK(ch == 2 && nf == 4096 && rs == 8 && rate == 48000, "cache WAV header %d %d %d %d", ch, nf, rs, rate);
        const double *d = (const double *)frames.data();
        double rest = 0;
        for (int i = 2; i < nf * ch; i++) rest = std::max(rest, std::fabs(d[i]));
        CHECK(std::fabs(d[0] - 1.0) < 1e-9 && std::fabs(d[1] - 1.0) < 1e-9 && rest < 1e-9, "flat EQ impulse %g %g rest %g", d[0], d[1], rest);
        remove(path.c_str());
    }
    // dither (dither.cpp) through both classes: 16-bit output with HP-TPDF dither stays within the dither's
    // reach of the undithered float result (y - x = e[n-1] - e[n-2] - e[n] with |e| <= 1.5 LSB), differs from the
    // undithered integers, and the class fills dither_state as the reference's constructor does
    {
        const int L = 256, B = 2, C = 2, nb = 6, srate = 300;
        std::mt19937 rng(11); std::uniform_real_distribution<float> u(-0.5f, 0.5f);
        std::vector<float> h0(300), h1(300), x((size_t)nb * L * C), yf(x.size());
        for (int n = 0; n < 300; n++) { h0[n] = u(rng) * 0.01f; h1[n] = u(rng) * 0.01f; }
        h0[0] = 0.7f; h1[0] = -0.6f;
        for (auto &v : x) v = u(rng);
        void *hp[2] = {h0.data(), h1.data()};
        std::vector<int16_t> yd(x.size()), yn(x.size());
        brutefir ff(L, B, 4, C, BF_SAMPLE_FORMAT_FLOAT_LE, BF_SAMPLE_FORMAT_FLOAT_LE, srate, false);
        brutefir fd(L, B, 4, C, BF_SAMPLE_FORMAT_FLOAT_LE, BFIR_SAMPLE_FORMAT_S16_LE, srate, true);
        brutefir fn(L, B, 4, C, BF_SAMPLE_FORMAT_FLOAT_LE, BFIR_SAMPLE_FORMAT_S16_LE, srate, false);
        CHECK(fd.create_error() == 0, "dither engine: %s", bfir_strerror(fd.create_error()));
        ff.set_coeff(hp, C, 300, B, 1.0); fd.set_coeff(hp, C, 300, B, 1.0); fn.set_coeff(hp, C, 300, B, 1.0);
        CHECK(ff.run_blocks(x.data(), yf.data(), nb) == 0 && fd.run_blocks(x.data(), yd.data(), nb) == 0 &&
              fn.run_blocks(x.data(), yn.data(), nb) == 0, "dither runs");
        double worst = 0; long differ = 0;
        for (size_t i = 0; i < x.size(); i++) {
            worst = std::max(worst, std::fabs((double)yd[i] - 32768.0 * yf[i]));
            differ += yd[i] != yn[i];
        }
        CHECK(worst <= 4.6, "dithered output strays %.2f LSB from the float result", worst);
        CHECK(differ > (long)x.size() / 10, "dither changed only %ld of %zu samples", differ, x.size());
        dither_state_t st[BF_MAXCHANNELS];
        dither dth(C, srate, 4, 0, L, st);
        CHECK(st[0].randtab_ptr == 1 && st[1].randtab_ptr == 10 * srate + 1, "dither_state %d %d", st[0].randtab_ptr, st[1].randtab_ptr);
        // the stage form: cbuf2raw with apply_dither on the same float block the engine produced
        fftw_convolver conv(L, 4, &dth);
        std::vector<float> cbuf(2 * L);
        std::vector<int16_t> raw((size_t)L * C, 0);
        buffer_format_t bf; bfoverflow_t of = {0, 0, 0.0, 32767.0};
        bf.sf.isfloat = false; bf.sf.swap = false; bf.sf.bytes = bf.sf.sbytes = 2; bf.sf.scale = 1.0; bf.sf.format = BFIR_SAMPLE_FORMAT_S16_LE;
        bf.sample_spacing = C; bf.byte_offset = 0;
        for (int n = 0; n < L; n++) cbuf[n] = 32768.0f * yf[(size_t)n * C];
        conv.convolver_cbuf2raw(cbuf.data(), raw.data(), &bf, true, &st[0], &of);
        CHECK(conv.last_status() == 0, "cbuf2raw with dither: %s", bfir_strerror(conv.last_status()));
        CHECK(st[0].randtab_ptr == 1 + L, "preloop must advance the table position: %d", st[0].randtab_ptr);
        long same = 0;
        for (int n = 0; n < L; n++) same += raw[(size_t)n * C] == yd[(size_t)n * C];
        CHECK(same == L, "stage and engine dither disagree on %ld of %d samples", L - same, L);
        fftw_convolver nod(L, 4, nullptr);
        nod.convolver_cbuf2raw(cbuf.data(), raw.data(), &bf, true, &st[0], &of);
        CHECK(nod.last_status() != 0, "cbuf2raw with dither but no dither instance must fail");
        // create_fft_plan hands out a real plan (fftw_convolver.cpp:653-675): R2HC then HC2R is n times the input
        void *fw = conv.create_fft_plan(9, 0, 0), *bw = conv.create_fft_plan(9, 1, 0);
        CHECK(fw && bw && fw != bw && conv.create_fft_plan(9, 0, 0) == fw, "create_fft_plan");
        std::vector<float> a(512), sp(512), b2(512);
        for (auto &v : a) v = u(rng);
        CHECK(bfir_fft_plan_execute((bfir_fft_plan *)fw, a.data(), sp.data()) == 0 &&
              bfir_fft_plan_execute((bfir_fft_plan *)bw, sp.data(), b2.data()) == 0, "plan execute");
        double e = 0;
        for (int n = 0; n < 512; n++) e = std::max(e, std::fabs((double)b2[n] / 512.0 - a[n]));
        CHECK(e < 1e-5, "plan round trip %g", e);
        conv.destroy_fft_plan(9, 0, 0);
    }
    // the td convolver the way delay::subsample uses it (delay.cpp:150-180): blocks of `blocklen` samples,
    // [rest | block] in, the first half out -- a filter whose only tap is 1 at `centre` delays the signal by
    // `centre` samples -- and the text dump of a coefficient block (fftw_convolver.cpp:604-651)
    {
        const int n_taps = 31, centre = n_taps >> 1, frag = 512;
        fftw_convolver conv(256, 8, nullptr);
        const int bl = conv.convolver_td_block_length(n_taps);
        CHECK(bl == 32 && conv.convolver_td_block_length(0) == -1 && conv.convolver_td_new(nullptr, 0) == nullptr, "td block length %d", bl);
        std::vector<double> filt(n_taps, 0.0), sig(frag), out(frag), rest(bl, 0.0), cb(2 * bl);
        filt[centre] = 1.0;
        std::mt19937 rng(5); std::uniform_real_distribution<double> u(-1.0, 1.0);
        for (auto &v : sig) v = u(rng);
        td_conv_t *tdc = conv.convolver_td_new(filt.data(), n_taps);
        CHECK(tdc && tdc->blocklen == bl && tdc->coeffs && tdc->fftplan && tdc->ifftplan, "convolver_td_new");
        for (int i = 0; tdc && i < frag; i += bl) {
            std::copy(rest.begin(), rest.end(), cb.begin());
            std::copy(sig.begin() + i, sig.begin() + i + bl, cb.begin() + bl);
            std::copy(cb.begin() + bl, cb.end(), rest.begin());
            conv.convolver_td_convolve(tdc, cb.data());
            CHECK(conv.last_status() == 0, "td_convolve: %s", bfir_strerror(conv.last_status()));
            std::copy(cb.begin(), cb.begin() + bl, out.begin() + i);
        }
        // tap k sits at lag bl + k of the circular product, so the first half of [rest | block] comes back as the
        // signal `centre` samples late: out[m] = sig[m - centre]
        double worst = 0;
        for (int m = 0; m < frag; m++) worst = std::max(worst, std::fabs(out[m] - (m < centre ? 0.0 : sig[m - centre])));
        CHECK(worst < 1e-12, "td delay line off by %g", worst);
        conv.convolver_td_free(tdc);
        std::vector<double> taps(256);
        for (auto &v : taps) v = u(rng);
        void *blk = conv.convolver_coeffs2cbuf(taps.data(), 256, 1.0, nullptr);
        const char *dir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
        std::string path = std::string(dir) + "/bfir_dump_test.txt";
        void *blks[2] = {blk, blk};
        conv.convolver_debug_dump_cbuf(path.c_str(), blks, 2);
        CHECK(conv.last_status() == 0, "debug_dump_cbuf: %s", bfir_strerror(conv.last_status()));
        FILE *fp = fopen(path.c_str(), "r");
        int lines = 0; double v, werr = 0;
        while (fp && fscanf(fp, "%lf", &v) == 1) { werr = std::max(werr, std::fabs(v - taps[lines % 256])); lines++; }
        if (fp) fclose(fp);
        CHECK(lines == 512 && werr < 1e-12, "dump: %d lines, error %g", lines, werr);
        remove(path.c_str());
        conv.convolver_debug_dump_cbuf("/nonexistent-dir/x.txt", blks, 2);
        CHECK(conv.last_status() == BFIR_ERR_IO, "unwritable dump file must report BFIR_ERR_IO");
        bfir_aligned_free(blk);
    }
    printf(g_fail ? "FAILED (%d)\n" : "ALL OK\n", g_fail);
    return g_fail ? 1 : 0;
}
