// mini_xml.h — a small XML reader sufficient for .scene files: elements,
// attributes (single or double quoted), comments, processing instructions and
// self-closing tags.  Text content is skipped; entities are not expanded
// (scene files carry numbers and file names only).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace tk {

struct XmlNode {
    std::string name;
    std::map<std::string, std::string> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;
    int line = 0;

    const XmlNode* first(const char* child) const {
        for (auto& c : children)
            if (c->name == child) return c.get();
        return nullptr;
    }
    std::vector<const XmlNode*> all(const char* child) const {
        std::vector<const XmlNode*> r;
        for (auto& c : children)
            if (c->name == child) r.push_back(c.get());
        return r;
    }
    const std::string* attr(const char* key) const {
        auto it = attrs.find(key);
        return it == attrs.end() ? nullptr : &it->second;
    }
};

class XmlParser {
public:
    explicit XmlParser(const std::string& text) : s(text) {}

    std::unique_ptr<XmlNode> parseDocument() {
        skipMisc();
        auto root = parseElement();
        if (!root) fail("no root element");
        return root;
    }

private:
    const std::string& s;
    size_t i = 0;
    int line = 1;

    [[noreturn]] void fail(const std::string& what) const {
        throw std::runtime_error("xml:" + std::to_string(line) + ": " + what);
    }
    bool eof() const { return i >= s.size(); }
    char peek() const { return eof() ? '\0' : s[i]; }
    char get() {
        char c = s[i++];
        if (c == '\n') ++line;
        return c;
    }
    bool startsWith(const char* lit) const { return s.compare(i, std::char_traits<char>::length(lit), lit) == 0; }
    void skipSpace() {
        while (!eof() && (peek() == ' ' || peek() == '\t' || peek() == '\r' || peek() == '\n')) get();
    }
    void skipUntil(const char* lit) {
        size_t n = std::char_traits<char>::length(lit);
        while (!eof() && s.compare(i, n, lit) != 0) get();
        if (eof()) fail(std::string("unterminated, expected ") + lit);
        for (size_t k = 0; k < n; ++k) get();
    }
    // whitespace, comments, <?...?>, <!DOCTYPE ...>, stray text
    void skipMisc() {
        for (;;) {
            while (!eof() && peek() != '<') get();
            if (eof()) return;
            if (startsWith("<!--")) skipUntil("-->");
            else if (startsWith("<?")) skipUntil("?>");
            else if (startsWith("<!")) skipUntil(">");
            else return;
        }
    }
    std::string parseName() {
        size_t b = i;
        while (!eof()) {
            char c = peek();
            if (c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '=' || c == '>' || c == '/') break;
            get();
        }
        if (i == b) fail("expected a name");
        return s.substr(b, i - b);
    }
    std::unique_ptr<XmlNode> parseElement() {
        if (peek() != '<') return nullptr;
        get();
        auto node = std::make_unique<XmlNode>();
        node->line = line;
        node->name = parseName();
        for (;;) {
            skipSpace();
            if (eof()) fail("unterminated tag <" + node->name);
            if (peek() == '/') {
                get();
                if (get() != '>') fail("expected '>' after '/'");
                return node;
            }
            if (peek() == '>') {
                get();
                break;
            }
            std::string key = parseName();
            skipSpace();
            if (get() != '=') fail("expected '=' after attribute " + key);
            skipSpace();
            char q = get();
            if (q != '"' && q != '\'') fail("attribute value must be quoted");
            size_t b = i;
            while (!eof() && peek() != q) get();
            if (eof()) fail("unterminated attribute value");
            node->attrs[key] = s.substr(b, i - b);
            get();
        }
        for (;;) {
            skipMisc();
            if (eof()) fail("missing </" + node->name + ">");
            if (startsWith("</")) {
                get();
                get();
                std::string close = parseName();
                skipSpace();
                if (get() != '>') fail("malformed closing tag");
                if (close != node->name) fail("</" + close + "> closes <" + node->name + ">");
                return node;
            }
            node->children.push_back(parseElement());
        }
    }
};

}  // namespace tk
