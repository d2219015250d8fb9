// A reader for TOML 1.0 documents, as far as configuration files of the ray tracer go: comments, bare / quoted /
// dotted keys, basic and literal strings (single- and multi-line, with escapes), integers (decimal with
// underscores, 0x / 0o / 0b), floats (fraction, exponent, inf, nan), booleans, arrays (nested, multi-line, trailing
// comma), inline tables, [tables] and [[arrays of tables]].  Dates and times are refused.  The reference reads its
// configuration with toml++ 3.2 (reference src/config.cpp:197); this is what that file needs of it.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rtamd::toml {

struct Value;
struct Table {
    std::vector<std::pair<std::string, Value>> entries; // in file order
    const Value* find(const std::string& key) const;
    Value* find(const std::string& key);
    Value& add(const std::string& key);
};

struct Value {
    enum Kind { None, Bool, Int, Float, String, Array, TableKind };
    Kind kind = None;
    bool boolean = false;
    int64_t integer = 0;
    double real = 0.0;
    std::string string;
    std::vector<Value> array;
    std::shared_ptr<Table> table;
    bool fixed = false;     // inline table / array literal: may not be extended by later headers
    bool defined = false;   // table named by a [header] already (a second one is an error)
    bool fromHeader = false;

    bool is_number() const { return kind == Int || kind == Float; }
    double number() const { return kind == Int ? double(integer) : real; }
    // the value under a key of this table (a None value when this is no table or has no such key)
    const Value& operator[](const std::string& key) const
    {
        static const Value none;
        if (kind != TableKind || !table) return none;
        const Value* v = table->find(key);
        return v ? *v : none;
    }
    explicit operator bool() const { return kind != None; }
};

inline const Value* Table::find(const std::string& key) const
{
    for (const auto& e : entries)
        if (e.first == key) return &e.second;
    return nullptr;
}
inline Value* Table::find(const std::string& key)
{
    for (auto& e : entries)
        if (e.first == key) return &e.second;
    return nullptr;
}
inline Value& Table::add(const std::string& key)
{
    entries.emplace_back(key, Value {});
    return entries.back().second;
}

class Parser {
public:
    explicit Parser(const std::string& text) : m_text(text) {}

    Value parse()
    {
        Value root = make_table();
        Table* current = root.table.get();
        while (true) {
            skip_blank_lines();
            if (at_end()) break;
            if (peek() == '[') {
                const bool arrayOfTables = m_text.compare(m_pos, 2, "[[") == 0;
                m_pos += arrayOfTables ? 2 : 1;
                skip_spaces();
                const std::vector<std::string> path = key_path();
                skip_spaces();
                expect(']');
                if (arrayOfTables) expect(']');
                end_of_line();
                current = open_header(root, path, arrayOfTables);
            } else {
                key_value(*current);
                end_of_line();
            }
        }
        return root;
    }

private:
    const std::string& m_text;
    size_t m_pos = 0;
    int m_depth = 0;
    int m_line = 1;

    [[noreturn]] void error(const std::string& what) const { throw std::runtime_error("TOML, line " + std::to_string(m_line) + ": " + what); }
    bool at_end() const { return m_pos >= m_text.size(); }
    char peek(size_t ahead = 0) const { return m_pos + ahead < m_text.size() ? m_text[m_pos + ahead] : '\0'; }
    void expect(char c)
    {
        if (peek() != c) error(std::string("expected '") + c + "'");
        m_pos++;
    }
    static Value make_table()
    {
        Value v;
        v.kind = Value::TableKind;
        v.table = std::make_shared<Table>();
        return v;
    }
    void skip_spaces()
    {
        while (peek() == ' ' || peek() == '\t') m_pos++;
    }
    void skip_comment()
    {
        if (peek() != '#') return;
        while (!at_end() && peek() != '\n') {
            const unsigned char c = (unsigned char)peek();
            if ((c < 0x20 && c != '\t' && c != '\r') || c == 0x7f) error("control character in comment");
            m_pos++;
        }
    }
    bool newline()
    {
        if (peek() == '\n') {
            m_pos++;
            m_line++;
            return true;
        }
        if (peek() == '\r' && peek(1) == '\n') {
            m_pos += 2;
            m_line++;
            return true;
        }
        return false;
    }
    void skip_blank_lines()
    {
        while (true) {
            skip_spaces();
            skip_comment();
            if (!newline()) return;
        }
    }
    void end_of_line()
    {
        skip_spaces();
        skip_comment();
        if (at_end()) return;
        if (!newline()) error("unexpected text after a value");
    }
    // white space, comments and line breaks inside arrays
    void skip_filler()
    {
        while (true) {
            skip_spaces();
            skip_comment();
            if (!newline()) return;
        }
    }

    static bool bare_key_char(char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_' || c == '-'; }

    std::vector<std::string> key_path()
    {
        std::vector<std::string> path;
        while (true) {
            skip_spaces();
            if (peek() == '"') {
                path.push_back(basic_string(false));
            } else if (peek() == '\'') {
                path.push_back(literal_string(false));
            } else {
                std::string k;
                while (bare_key_char(peek())) k.push_back(m_text[m_pos++]);
                if (k.empty()) error("expected a key");
                path.push_back(k);
            }
            skip_spaces();
            if (peek() != '.') return path;
            m_pos++;
        }
    }

    // the table a dotted key or header path leads to, creating what is missing on the way
    Table* descend(Table* t, const std::string& key, bool viaHeader)
    {
        Value* v = t->find(key);
        if (!v) {
            Value& made = t->add(key);
            made = make_table();
            made.fromHeader = viaHeader;
            return made.table.get();
        }
        if (v->kind == Value::TableKind) {
            if (v->fixed) error("'" + key + "' is an inline table and cannot be extended");
            if (!viaHeader && v->fromHeader && v->defined) error("'" + key + "' was defined by a table header");
            return v->table.get();
        }
        if (v->kind == Value::Array && viaHeader && !v->fixed && !v->array.empty() && v->array.back().kind == Value::TableKind) return v->array.back().table.get();
        error("'" + key + "' is not a table");
    }

    Table* open_header(Value& root, const std::vector<std::string>& path, bool arrayOfTables)
    {
        Table* t = root.table.get();
        for (size_t i = 0; i + 1 < path.size(); i++) t = descend(t, path[i], true);
        const std::string& last = path.back();
        Value* v = t->find(last);
        if (arrayOfTables) {
            if (!v) {
                Value& made = t->add(last);
                made.kind = Value::Array;
                v = &made;
            } else if (v->kind != Value::Array || v->fixed) {
                error("'" + last + "' is not an array of tables");
            }
            v->array.push_back(make_table());
            return v->array.back().table.get();
        }
        if (!v) {
            Value& made = t->add(last);
            made = make_table();
            made.defined = true;
            made.fromHeader = true;
            return made.table.get();
        }
        if (v->kind != Value::TableKind || v->fixed || v->defined) error("table '" + last + "' defined twice");
        v->defined = true;
        v->fromHeader = true;
        return v->table.get();
    }

    void key_value(Table& into)
    {
        const std::vector<std::string> path = key_path();
        skip_spaces();
        expect('=');
        skip_spaces();
        Table* t = &into;
        for (size_t i = 0; i + 1 < path.size(); i++) t = descend(t, path[i], false);
        if (t->find(path.back())) error("key '" + path.back() + "' defined twice");
        Value v = value();
        t->add(path.back()) = std::move(v);
    }

    // arrays and inline tables nest through value(); a hostile file must not exhaust the stack
    enum { MAX_NESTING = 64 };
    struct Nesting {
        int& depth;
        explicit Nesting(int& d) : depth(d) { depth++; }
        ~Nesting() { depth--; }
    };

    Value value()
    {
        const Nesting nesting(m_depth);
        if (m_depth > MAX_NESTING) error("arrays / inline tables nested deeper than 64 levels");
        const char c = peek();
        Value v;
        if (c == '"') {
            v.kind = Value::String;
            v.string = basic_string(true);
        } else if (c == '\'') {
            v.kind = Value::String;
            v.string = literal_string(true);
        } else if (c == '[') {
            m_pos++;
            v.kind = Value::Array;
            v.fixed = true;
            while (true) {
                skip_filler();
                if (peek() == ']') break;
                v.array.push_back(value());
                skip_filler();
                if (peek() == ',') {
                    m_pos++;
                    continue;
                }
                if (peek() != ']') error("expected ',' or ']' in array");
            }
            m_pos++;
        } else if (c == '{') {
            m_pos++;
            v = make_table();
            skip_spaces();
            if (peek() != '}') {
                while (true) {
                    skip_spaces();
                    key_value(*v.table);
                    skip_spaces();
                    if (peek() == ',') {
                        m_pos++;
                        continue;
                    }
                    break;
                }
            }
            expect('}');
            fix(v);
        } else if (m_text.compare(m_pos, 4, "true") == 0) {
            m_pos += 4;
            v.kind = Value::Bool;
            v.boolean = true;
        } else if (m_text.compare(m_pos, 5, "false") == 0) {
            m_pos += 5;
            v.kind = Value::Bool;
        } else {
            v = number();
        }
        return v;
    }
    static void fix(Value& v)
    {
        v.fixed = true;
        if (v.kind == Value::TableKind)
            for (auto& e : v.table->entries) fix(e.second);
        for (Value& e : v.array) fix(e);
    }

    Value number()
    {
        size_t end = m_pos;
        while (end < m_text.size()) {
            const char c = m_text[end];
            if ((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '+' || c == '-' || c == '.' || c == '_' || c == ':') end++;
            else break;
        }
        std::string tok = m_text.substr(m_pos, end - m_pos);
        if (tok.empty()) error("expected a value");
        m_pos = end;
        Value v;
        std::string body = tok;
        int sign = 1;
        if (body[0] == '+' || body[0] == '-') {
            sign = body[0] == '-' ? -1 : 1;
            body.erase(0, 1);
        }
        if (body == "inf" || body == "nan") {
            v.kind = Value::Float;
            v.real = body == "inf" ? sign * INFINITY : std::copysign(NAN, double(sign));
            return v;
        }
        if (body.empty() || !(body[0] >= '0' && body[0] <= '9')) error("unknown value '" + tok + "'");
        auto strip = [&](const std::string& digits, auto valid) {
            std::string out;
            for (size_t i = 0; i < digits.size(); i++) {
                if (digits[i] == '_') {
                    if (i == 0 || i + 1 == digits.size() || !valid(digits[i - 1]) || !valid(digits[i + 1])) error("misplaced '_' in '" + tok + "'");
                    continue;
                }
                if (!valid(digits[i])) error("bad digit in '" + tok + "'");
                out.push_back(digits[i]);
            }
            if (out.empty()) error("no digits in '" + tok + "'");
            return out;
        };
        if (body.size() > 2 && body[0] == '0' && (body[1] == 'x' || body[1] == 'o' || body[1] == 'b')) {
            if (tok[0] == '+' || tok[0] == '-') error("sign on a prefixed integer");
            const int base = body[1] == 'x' ? 16 : (body[1] == 'o' ? 8 : 2);
            const std::string digits = strip(body.substr(2), [base](char c) {
                if (base == 16) return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F');
                return c >= '0' && c < '0' + base;
            });
            v.kind = Value::Int;
            v.integer = int64_t(std::stoull(digits, nullptr, base));
            return v;
        }
        auto isDigit = [](char c) { return c >= '0' && c <= '9'; };
        if (body.find(':') != std::string::npos || (body.size() >= 10 && body[4] == '-' && body[7] == '-')) error("dates and times are not supported");
        const size_t mark = body.find_first_of(".eE");
        if (mark == std::string::npos) {
            const std::string digits = strip(body, isDigit);
            if (digits.size() > 1 && digits[0] == '0') error("leading zero in '" + tok + "'");
            v.kind = Value::Int;
            try {
                v.integer = std::stoll((sign < 0 ? "-" : "") + digits);
            } catch (const std::exception&) {
                error("integer out of range: '" + tok + "'");
            }
            return v;
        }
        // float: integer part [. fraction] [e exponent]
        std::string whole = body.substr(0, mark), fraction, exponent;
        size_t at = mark;
        if (body[at] == '.') {
            const size_t e = body.find_first_of("eE", at);
            fraction = body.substr(at + 1, e == std::string::npos ? std::string::npos : e - at - 1);
            if (fraction.empty()) error("no digits after '.' in '" + tok + "'");
            at = e;
        }
        if (at != std::string::npos) {
            exponent = body.substr(at + 1);
            if (exponent.empty()) error("empty exponent in '" + tok + "'");
        }
        whole = strip(whole, isDigit);
        if (whole.size() > 1 && whole[0] == '0') error("leading zero in '" + tok + "'");
        std::string text = (sign < 0 ? "-" : "") + whole;
        if (!fraction.empty()) text += "." + strip(fraction, isDigit);
        if (!exponent.empty()) {
            std::string esign;
            if (exponent[0] == '+' || exponent[0] == '-') {
                esign = exponent.substr(0, 1);
                exponent.erase(0, 1);
            }
            text += "e" + esign + strip(exponent, isDigit);
        }
        v.kind = Value::Float;
        v.real = std::strtod(text.c_str(), nullptr);
        return v;
    }

    void append_utf8(std::string& out, uint32_t cp)
    {
        if (cp >= 0xd800 && cp <= 0xdfff) error("escape is not a scalar value");
        if (cp > 0x10ffff) error("escape is not a scalar value");
        if (cp < 0x80) out.push_back(char(cp));
        else if (cp < 0x800) {
            out.push_back(char(0xc0 | (cp >> 6)));
            out.push_back(char(0x80 | (cp & 0x3f)));
        } else if (cp < 0x10000) {
            out.push_back(char(0xe0 | (cp >> 12)));
            out.push_back(char(0x80 | ((cp >> 6) & 0x3f)));
            out.push_back(char(0x80 | (cp & 0x3f)));
        } else {
            out.push_back(char(0xf0 | (cp >> 18)));
            out.push_back(char(0x80 | ((cp >> 12) & 0x3f)));
            out.push_back(char(0x80 | ((cp >> 6) & 0x3f)));
            out.push_back(char(0x80 | (cp & 0x3f)));
        }
    }

    std::string basic_string(bool allowMultiline)
    {
        if (allowMultiline && m_text.compare(m_pos, 3, "\"\"\"") == 0) {
            m_pos += 3;
            newline(); // a line break right after the opening quotes is dropped
            std::string out;
            while (true) {
                if (at_end()) error("unterminated string");
                if (m_text.compare(m_pos, 3, "\"\"\"") == 0) {
                    m_pos += 3;
                    // up to two more quotes belong to the text
                    for (int extra = 0; extra < 2 && peek() == '"'; extra++) {
                        out.push_back('"');
                        m_pos++;
                    }
                    return out;
                }
                if (peek() == '\\') {
                    // a backslash at the end of a line swallows the break and the white space that follows
                    size_t p = m_pos + 1;
                    while (p < m_text.size() && (m_text[p] == ' ' || m_text[p] == '\t')) p++;
                    if (p < m_text.size() && (m_text[p] == '\n' || (m_text[p] == '\r' && p + 1 < m_text.size() && m_text[p + 1] == '\n'))) {
                        m_pos = p;
                        while (true) {
                            skip_spaces();
                            if (!newline()) break;
                        }
                        continue;
                    }
                    escape(out);
                    continue;
                }
                if (newline()) {
                    out.push_back('\n');
                    continue;
                }
                plain_char(out);
            }
        }
        expect('"');
        std::string out;
        while (true) {
            if (at_end() || peek() == '\n') error("unterminated string");
            if (peek() == '"') {
                m_pos++;
                return out;
            }
            if (peek() == '\\') escape(out);
            else plain_char(out);
        }
    }
    void plain_char(std::string& out)
    {
        const unsigned char c = (unsigned char)peek();
        if ((c < 0x20 && c != '\t') || c == 0x7f) error("control character in string");
        out.push_back(char(c));
        m_pos++;
    }
    void escape(std::string& out)
    {
        m_pos++; // the backslash
        const char c = peek();
        m_pos++;
        switch (c) {
        case 'b': out.push_back('\b'); return;
        case 't': out.push_back('\t'); return;
        case 'n': out.push_back('\n'); return;
        case 'f': out.push_back('\f'); return;
        case 'r': out.push_back('\r'); return;
        case '"': out.push_back('"'); return;
        case '\\': out.push_back('\\'); return;
        case 'u':
        case 'U': {
            const int n = c == 'u' ? 4 : 8;
            uint32_t cp = 0;
            for (int i = 0; i < n; i++) {
                const char h = peek();
                int d;
                if (h >= '0' && h <= '9') d = h - '0';
                else if (h >= 'a' && h <= 'f') d = h - 'a' + 10;
                else if (h >= 'A' && h <= 'F') d = h - 'A' + 10;
                else error("bad unicode escape");
                cp = cp * 16 + uint32_t(d);
                m_pos++;
            }
            append_utf8(out, cp);
            return;
        }
        default: error("unknown escape");
        }
    }
    std::string literal_string(bool allowMultiline)
    {
        if (allowMultiline && m_text.compare(m_pos, 3, "'''") == 0) {
            m_pos += 3;
            newline();
            std::string out;
            while (true) {
                if (at_end()) error("unterminated string");
                if (m_text.compare(m_pos, 3, "'''") == 0) {
                    m_pos += 3;
                    for (int extra = 0; extra < 2 && peek() == '\''; extra++) {
                        out.push_back('\'');
                        m_pos++;
                    }
                    return out;
                }
                if (newline()) {
                    out.push_back('\n');
                    continue;
                }
                plain_char(out);
            }
        }
        expect('\'');
        std::string out;
        while (true) {
            if (at_end() || peek() == '\n') error("unterminated string");
            if (peek() == '\'') {
                m_pos++;
                return out;
            }
            plain_char(out);
        }
    }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

} // namespace rtamd::toml
