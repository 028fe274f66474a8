// Navigation-bit integrator, host side.
//
// Consumes the 1 kHz pseudosymbol stream a tracking channel emits and produces 50 bit/s navigation bits with the
// decisions of gypsum/navigation_bit_intergrator.py:100-288 (bit-phase selection over the last 320 pseudosymbols,
// periodic / health-triggered resynchronisation gated on receiver time < 40 s, 20-symbol majority vote with the
// <= 50 % confidence -> UNKNOWN rule, 30-unknowns reset).  Pure host C++: at 200x real time x 12 channels x 64
// streams the reference's per-pseudosymbol Python objects dominate host time; this keeps one flat state per channel
// and is fed whole blocks of gyp_track_rec straight from gyp_track_block's output.
#pragma once

#include <cmath>
#include <cstdint>
#include <deque>
#include <vector>

namespace gyp_bits_impl {

constexpr int kSymbolsPerBit = 20;            // constants.py:25 PSEUDOSYMBOLS_PER_NAVIGATION_BIT
constexpr int kSymbolsPerSecond = 1000;       // constants.py:26
constexpr int kSeenHistory = 1000;            // navigation_bit_intergrator.py:85 deque(maxlen=1000)
constexpr int kBitHistory = 50;               // :87 deque(maxlen=BITS_PER_SECOND)
constexpr int kPhaseSelectMin = kSymbolsPerBit * 4;      // :105
constexpr int kPhaseSelectWindow = kSymbolsPerBit * 16;  // :136
constexpr int kResyncPeriod = kSymbolsPerSecond * 1;     // :106, config.py:40
constexpr int kHealthMemory = 10;             // config.py:43
constexpr double kHealthThresholdPercent = 50.0;  // config.py:45
constexpr double kResyncDeadlineSeconds = 40.0;   // :276
constexpr int kUnknownRunReset = 30;          // :170

enum : int32_t { kBitZero = 0, kBitOne = 1, kBitUnknown = 2 };

struct Symbol {
    double start, end;
    int8_t value;
};

struct BitOut {
    double start, end;
    int32_t bit;
};

struct Channel {
    // NavigationBitIntegratorHistory, :55-97
    int8_t seen[kSeenHistory];       // ring of the last 1000 pseudosymbol values
    int seen_len = 0, seen_head = 0; // head = index of the oldest entry
    int8_t bits[kBitHistory];
    int bits_len = 0, bits_head = 0;
    int32_t previous_bit_phase_decision = -1;   // -1 = None
    int32_t determined_bit_phase = -1;
    int64_t failed_bit_count = 0, emitted_bit_count = 0, processed = 0;
    int32_t sequential_unknown = 0;
    std::vector<Symbol> queue;       // queued_pseudosymbols
    int64_t cursor = 0;              // pseudosymbol_cursor_within_queue (can go negative, see slice_start)
    int64_t slide = 0;

    void reset() { *this = Channel(); }

    void seen_push(int8_t v) {
        if (seen_len < kSeenHistory) {
            seen[(seen_head + seen_len++) % kSeenHistory] = v;
        } else {
            seen[seen_head] = v;
            seen_head = (seen_head + 1) % kSeenHistory;
        }
    }
    void bits_push(int8_t v) {
        if (bits_len < kBitHistory) {
            bits[(bits_head + bits_len++) % kBitHistory] = v;
        } else {
            bits[bits_head] = v;
            bits_head = (bits_head + 1) % kBitHistory;
        }
    }

    // _redetermine_bit_phase, :129-147.  Returns -1 for None.
    int32_t redetermine() const {
        if (seen_len < kPhaseSelectMin) return -1;
        const int n = seen_len < kPhaseSelectWindow ? seen_len : kPhaseSelectWindow;
        int8_t w[kPhaseSelectWindow];
        for (int i = 0; i < n; ++i) w[i] = seen[(seen_head + seen_len - n + i) % kSeenHistory];
        const int full_bits = n / kSymbolsPerBit;   // chunks() drops the truncated tail, utils.py:28-38
        int32_t best = 0;
        double best_score = -1.0;
        for (int phase = 0; phase < kSymbolsPerBit; ++phase) {
            int64_t sum_abs = 0;
            for (int b = 0; b < full_bits; ++b) {
                int s = 0;
                for (int j = 0; j < kSymbolsPerBit; ++j) s += w[(phase + b * kSymbolsPerBit + j) % n];  // np.roll(-phase)
                sum_abs += s < 0 ? -s : s;
            }
            // same float64 operations, same order as :121-126
            double score = static_cast<double>(sum_abs) / (static_cast<double>(n) / kSymbolsPerBit);
            score = score / kSymbolsPerBit;
            if (score > best_score) {   // max(dict, key=dict.get): first maximum wins
                best_score = score;
                best = phase;
            }
        }
        return best;
    }

    // _should_resynchronize_bit_phase, :213-243
    bool should_resync() const {
        if (processed % kResyncPeriod == 0) return true;
        if (processed % kSymbolsPerBit != 0) return false;
        if (previous_bit_phase_decision < 0) return true;
        if (bits_len >= kHealthMemory) {
            int failed = 0;
            for (int i = 0; i < kHealthMemory; ++i)
                failed += bits[(bits_head + bits_len - kHealthMemory + i) % kBitHistory] == kBitUnknown;
            const double percent = (static_cast<double>(failed) / kHealthMemory) * 100.0;
            if (percent >= kHealthThresholdPercent) return true;
        }
        return false;
    }

    // _resynchronize_bit_phase_if_necessary, :245-277
    void resync_if_necessary() {
        if (!should_resync()) return;
        const int32_t prev = previous_bit_phase_decision;
        const int32_t now = redetermine();
        previous_bit_phase_decision = now;
        determined_bit_phase = now;
        if (prev < 0 && now >= 0) {
            if (now > 0) {
                cursor = now;
                slide = now;
            }
        } else if (prev >= 0 && now >= 0 && prev != now) {
            const int32_t diff = now - prev;
            slide += diff;
            cursor += diff;
        }
    }

    // _get_bit_value_from_pseudosymbols + _emit_bit_from_pseudosymbols, :149-193
    BitOut emit_bit(const Symbol* s) {
        int sum = 0;
        for (int j = 0; j < kSymbolsPerBit; ++j) sum += s[j].value;
        int32_t bit = sum > 0 ? kBitOne : kBitZero;
        const double scaled = (static_cast<double>(sum) / kSymbolsPerBit) * 100.0;
        const long confidence = std::labs(static_cast<long>(scaled));   // abs(int(...)): truncation toward zero
        if (confidence <= 50) bit = kBitUnknown;
        bits_push(static_cast<int8_t>(bit));
        if (bit == kBitUnknown) {
            ++sequential_unknown;
            ++failed_bit_count;
            if (sequential_unknown >= kUnknownRunReset) determined_bit_phase = -1;
        } else {
            sequential_unknown = 0;
        }
        return BitOut{s[0].start, s[kSymbolsPerBit - 1].end, bit};
    }

    // _emit_bits_from_queued_pseudosymbols, :195-211
    template <class Sink>
    void emit_from_queue(Sink&& sink) {
        if (determined_bit_phase < 0) return;
        const int64_t len = static_cast<int64_t>(queue.size());
        // Python slice semantics of queued[cursor:] (negative cursor counts from the end, both ends clamp)
        int64_t begin = cursor < 0 ? len + cursor : cursor;
        if (begin < 0) begin = 0;
        if (begin > len) begin = len;
        // the chunk list is materialised from the snapshot before any bit is emitted; emit_bit can clear
        // determined_bit_phase midway and the loop still finishes, as in the reference
        for (int64_t i = begin; len - i >= kSymbolsPerBit; i += kSymbolsPerBit) {
            sink(emit_bit(queue.data() + i));
            cursor += kSymbolsPerBit;
            ++emitted_bit_count;
        }
        if (len >= kSymbolsPerBit) {
            const int64_t offset_from_end = len - cursor;
            queue.erase(queue.begin(), queue.end() - kSymbolsPerBit);
            cursor = kSymbolsPerBit - offset_from_end;
        }
    }

    // process_pseudosymbol, :267-288.  Returns the slide recorded as cursor_at_emit_time.
    template <class Sink>
    int64_t process(double receiver_timestamp, const Symbol& sym, Sink&& sink) {
        const int64_t cursor_at_emit = slide;
        queue.push_back(sym);
        seen_push(sym.value);
        if (receiver_timestamp < kResyncDeadlineSeconds) resync_if_necessary();
        emit_from_queue(sink);
        ++processed;
        return cursor_at_emit;
    }
};

}  // namespace gyp_bits_impl
