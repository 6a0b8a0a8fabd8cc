// xml_lite.h — minimal XML DOM reader for the Mitsuba-0.x scene subset.
// Own code (the reference vendors pugixml, src/3rdparty/pugixml.cpp; only the
// features parse_scene.cpp uses are provided: elements, attributes, comments,
// <?xml?> declarations, the five predefined entities).
#pragma once
#include <cctype>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace gdpt {

struct XmlNode {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;

    bool has_attr(const std::string &key) const {
        for (auto &a : attrs) if (a.first == key) return true;
        return false;
    }
    // pugixml's attribute(...).value() returns "" when absent; parse_scene.cpp relies on that.
    std::string attr(const std::string &key) const {
        for (auto &a : attrs) if (a.first == key) return a.second;
        return "";
    }
    const XmlNode *child(const std::string &n) const {
        for (auto &c : children) if (c->name == n) return c.get();
        return nullptr;
    }
};

class XmlParser {
public:
    explicit XmlParser(const std::string &text) : s(text), i(0) {}

    std::unique_ptr<XmlNode> parse_document() {
        auto root = std::make_unique<XmlNode>();
        root->name = "#document";
        skip_misc();
        while (i < s.size()) {
            if (s[i] != '<') fail("text outside of the root element");
            root->children.push_back(parse_element());
            skip_misc();
        }
        return root;
    }

private:
    const std::string &s;
    size_t i;

    [[noreturn]] void fail(const std::string &msg) const {
        throw std::runtime_error("XML parse error at offset " + std::to_string(i) + ": " + msg);
    }
    bool starts(const char *lit) const { return s.compare(i, std::char_traits<char>::length(lit), lit) == 0; }
    void skip_ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
    void skip_until(const char *lit) {
        size_t p = s.find(lit, i);
        if (p == std::string::npos) fail(std::string("unterminated construct, expected ") + lit);
        i = p + std::char_traits<char>::length(lit);
    }
    // whitespace, comments, processing instructions, DOCTYPE
    void skip_misc() {
        for (;;) {
            skip_ws();
            if (starts("<!--")) { i += 4; skip_until("-->"); }
            else if (starts("<?")) { i += 2; skip_until("?>"); }
            else if (starts("<!DOCTYPE")) { skip_until(">"); }
            else break;
        }
    }
    static bool name_char(char c) {
        return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.';
    }
    std::string parse_name() {
        size_t b = i;
        while (i < s.size() && name_char(s[i])) i++;
        if (b == i) fail("expected a name");
        return s.substr(b, i - b);
    }
    static std::string decode_entities(const std::string &v) {
        if (v.find('&') == std::string::npos) return v;
        std::string o;
        for (size_t k = 0; k < v.size();) {
            if (v[k] == '&') {
                auto try_ent = [&](const char *e, char c) {
                    size_t n = std::char_traits<char>::length(e);
                    if (v.compare(k, n, e) == 0) { o.push_back(c); k += n; return true; }
                    return false;
                };
                if (try_ent("&amp;", '&') || try_ent("&lt;", '<') || try_ent("&gt;", '>') ||
                    try_ent("&quot;", '"') || try_ent("&apos;", '\'')) continue;
            }
            o.push_back(v[k++]);
        }
        return o;
    }
    std::unique_ptr<XmlNode> parse_element() {
        if (s[i] != '<') fail("expected '<'");
        i++;
        auto node = std::make_unique<XmlNode>();
        node->name = parse_name();
        for (;;) {
            skip_ws();
            if (i >= s.size()) fail("unterminated start tag");
            if (s[i] == '/') {
                if (i + 1 >= s.size() || s[i + 1] != '>') fail("expected '/>'");
                i += 2;
                return node;
            }
            if (s[i] == '>') { i++; break; }
            std::string key = parse_name();
            skip_ws();
            if (i >= s.size() || s[i] != '=') fail("expected '=' after attribute name");
            i++;
            skip_ws();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) fail("expected a quoted attribute value");
            char q = s[i++];
            size_t e = s.find(q, i);
            if (e == std::string::npos) fail("unterminated attribute value");
            node->attrs.emplace_back(key, decode_entities(s.substr(i, e - i)));
            i = e + 1;
        }
        // content
        for (;;) {
            if (i >= s.size()) fail("unterminated element <" + node->name + ">");
            if (starts("<!--")) { i += 4; skip_until("-->"); continue; }
            if (starts("<![CDATA[")) { skip_until("]]>"); continue; }
            if (starts("<?")) { i += 2; skip_until("?>"); continue; }
            if (starts("</")) {
                i += 2;
                std::string close = parse_name();
                if (close != node->name) fail("mismatched close tag </" + close + "> for <" + node->name + ">");
                skip_ws();
                if (i >= s.size() || s[i] != '>') fail("expected '>'");
                i++;
                return node;
            }
            if (s[i] == '<') { node->children.push_back(parse_element()); continue; }
            i++; // character data is irrelevant for Mitsuba scenes
        }
    }
};

} // namespace gdpt
