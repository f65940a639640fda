// Test-infrastructure shim exposing the subset of the tinyxml2 API that the
// reference's src/xml/*.cc uses (tinyxml2 itself is a FetchContent dependency and is
// not on disk). Written from scratch for the oracle build: a small non-validating
// XML DOM parser with line numbers, and a pretty printer. Not a product component.
#ifndef ORACLE_STUB_TINYXML2_H_
#define ORACLE_STUB_TINYXML2_H_

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace tinyxml2 {

enum XMLError {
  XML_SUCCESS = 0,
  XML_NO_ATTRIBUTE,
  XML_WRONG_ATTRIBUTE_TYPE,
  XML_ERROR_FILE_NOT_FOUND,
  XML_ERROR_FILE_COULD_NOT_BE_OPENED,
  XML_ERROR_FILE_READ_ERROR,
  XML_ERROR_PARSING_ELEMENT,
  XML_ERROR_PARSING_ATTRIBUTE,
  XML_ERROR_PARSING_TEXT,
  XML_ERROR_PARSING_CDATA,
  XML_ERROR_PARSING_COMMENT,
  XML_ERROR_PARSING_DECLARATION,
  XML_ERROR_PARSING_UNKNOWN,
  XML_ERROR_EMPTY_DOCUMENT,
  XML_ERROR_MISMATCHED_ELEMENT,
  XML_ERROR_PARSING,
  XML_ERROR_COUNT
};

class XMLDocument;
class XMLElement;
class XMLComment;
class XMLText;
class XMLPrinter;

class XMLAttribute {
 public:
  const char* Name() const { return name_.c_str(); }
  const char* Value() const { return value_.c_str(); }
  const XMLAttribute* Next() const { return next_; }
  int GetLineNum() const { return line_; }

 private:
  friend class XMLElement;
  friend class XMLDocument;
  std::string name_, value_;
  XMLAttribute* next_ = nullptr;
  int line_ = 0;
};

class XMLNode {
 public:
  virtual ~XMLNode() {}
  const char* Value() const { return value_.c_str(); }
  void SetValue(const char* v) { value_ = v ? v : ""; }
  int GetLineNum() const { return line_; }
  XMLDocument* GetDocument() const { return doc_; }
  XMLNode* Parent() const { return parent_; }
  bool NoChildren() const { return first_ == nullptr; }
  XMLNode* FirstChild() const { return first_; }
  XMLNode* LastChild() const { return last_; }
  XMLNode* NextSibling() const { return next_; }
  XMLNode* PreviousSibling() const { return prev_; }

  virtual XMLElement* ToElement() { return nullptr; }
  virtual XMLComment* ToComment() { return nullptr; }
  virtual XMLText* ToText() { return nullptr; }
  virtual XMLDocument* ToDocument() { return nullptr; }
  virtual const XMLElement* ToElement() const { return nullptr; }

  inline XMLElement* FirstChildElement(const char* name = nullptr) const;
  inline XMLElement* NextSiblingElement(const char* name = nullptr) const;

  XMLNode* InsertEndChild(XMLNode* n) {
    Unlink(n);
    n->parent_ = this;
    n->prev_ = last_;
    n->next_ = nullptr;
    if (last_) last_->next_ = n; else first_ = n;
    last_ = n;
    return n;
  }
  XMLNode* LinkEndChild(XMLNode* n) { return InsertEndChild(n); }
  XMLNode* InsertFirstChild(XMLNode* n) {
    Unlink(n);
    n->parent_ = this;
    n->prev_ = nullptr;
    n->next_ = first_;
    if (first_) first_->prev_ = n; else last_ = n;
    first_ = n;
    return n;
  }
  XMLNode* InsertAfterChild(XMLNode* after, XMLNode* n) {
    if (!after || after->parent_ != this) return nullptr;
    if (after == last_) return InsertEndChild(n);
    Unlink(n);
    n->parent_ = this;
    n->prev_ = after;
    n->next_ = after->next_;
    after->next_->prev_ = n;
    after->next_ = n;
    return n;
  }
  void DeleteChild(XMLNode* n) {
    if (n && n->parent_ == this) Unlink(n);  // storage stays owned by the document
  }
  void DeleteChildren() { while (first_) Unlink(first_); }

  inline XMLNode* DeepClone(XMLDocument* target) const;
  virtual XMLNode* ShallowClone(XMLDocument* target) const = 0;

 protected:
  friend class XMLDocument;
  friend class XMLElement;
  static void Unlink(XMLNode* n) {
    XMLNode* p = n->parent_;
    if (!p) return;
    if (n->prev_) n->prev_->next_ = n->next_; else p->first_ = n->next_;
    if (n->next_) n->next_->prev_ = n->prev_; else p->last_ = n->prev_;
    n->parent_ = n->prev_ = n->next_ = nullptr;
  }
  std::string value_;
  int line_ = 0;
  XMLDocument* doc_ = nullptr;
  XMLNode *parent_ = nullptr, *first_ = nullptr, *last_ = nullptr, *prev_ = nullptr, *next_ = nullptr;
};

class XMLComment : public XMLNode {
 public:
  XMLComment* ToComment() override { return this; }
  inline XMLNode* ShallowClone(XMLDocument* target) const override;
};

class XMLText : public XMLNode {
 public:
  XMLText* ToText() override { return this; }
  inline XMLNode* ShallowClone(XMLDocument* target) const override;
};

class XMLElement : public XMLNode {
 public:
  XMLElement* ToElement() override { return this; }
  const XMLElement* ToElement() const override { return this; }
  const char* Name() const { return Value(); }
  void SetName(const char* n) { SetValue(n); }

  const char* Attribute(const char* name, const char* value = nullptr) const {
    for (const XMLAttribute* a = attr_; a; a = a->next_) {
      if (a->name_ == name) {
        if (!value || a->value_ == value) return a->value_.c_str();
        return nullptr;
      }
    }
    return nullptr;
  }
  const XMLAttribute* FirstAttribute() const { return attr_; }
  const XMLAttribute* FindAttribute(const char* name) const {
    for (const XMLAttribute* a = attr_; a; a = a->next_) if (a->name_ == name) return a;
    return nullptr;
  }
  inline void SetAttribute(const char* name, const char* value);
  void SetAttribute(const char* name, int v) { SetAttribute(name, std::to_string(v).c_str()); }
  void SetAttribute(const char* name, unsigned v) { SetAttribute(name, std::to_string(v).c_str()); }
  void SetAttribute(const char* name, long long v) { SetAttribute(name, std::to_string(v).c_str()); }
  void SetAttribute(const char* name, bool v) { SetAttribute(name, v ? "true" : "false"); }
  void SetAttribute(const char* name, double v) {
    char buf[64];
    snprintf(buf, sizeof(buf), "%.17g", v);
    SetAttribute(name, buf);
  }
  void DeleteAttribute(const char* name) {
    XMLAttribute** pp = &attr_;
    while (*pp) {
      if ((*pp)->name_ == name) { *pp = (*pp)->next_; return; }
      pp = &(*pp)->next_;
    }
  }
  const char* GetText() const {
    for (XMLNode* c = first_; c; c = c->NextSibling()) if (c->ToText()) return c->Value();
    return nullptr;
  }
  inline void SetText(const char* text);
  inline XMLNode* ShallowClone(XMLDocument* target) const override;

 private:
  friend class XMLDocument;
  friend class XMLPrinter;
  XMLAttribute* attr_ = nullptr;
};

class XMLPrinter {
 public:
  XMLPrinter(FILE* file = nullptr, bool compact = false, int depth = 0)
      : file_(file), compact_(compact), depth_(depth) {}
  virtual ~XMLPrinter() {}
  const char* CStr() const { return buf_.c_str(); }
  int CStrSize() const { return (int)buf_.size() + 1; }
  void ClearBuffer() { buf_.clear(); }
  virtual void PrintSpace(int depth) { for (int i = 0; i < depth; ++i) Write("    "); }
  void Write(const char* s) { Write(s, strlen(s)); }
  void Write(const char* s, size_t n) {
    if (file_) fwrite(s, 1, n, file_); else buf_.append(s, n);
  }
  inline void PrintNode(const XMLNode* n, int depth);

 private:
  void WriteEscaped(const std::string& s, bool attr) {
    for (char c : s) {
      switch (c) {
        case '&': Write("&amp;"); break;
        case '<': Write("&lt;"); break;
        case '>': Write("&gt;"); break;
        case '"': if (attr) Write("&quot;"); else Write("\""); break;
        default: Write(&c, 1);
      }
    }
  }
  FILE* file_;
  bool compact_;
  int depth_;
  std::string buf_;
};

class XMLDocument : public XMLNode {
 public:
  XMLDocument() { doc_ = this; }
  XMLDocument(const XMLDocument&) = delete;
  XMLDocument& operator=(const XMLDocument&) = delete;
  XMLDocument* ToDocument() override { return this; }
  XMLNode* ShallowClone(XMLDocument*) const override { return nullptr; }

  XMLElement* RootElement() const { return FirstChildElement(); }
  bool Error() const { return err_ != XML_SUCCESS; }
  XMLError ErrorID() const { return err_; }
  const char* ErrorStr() const { return errstr_.c_str(); }
  int ErrorLineNum() const { return errline_; }
  void ClearError() { err_ = XML_SUCCESS; errstr_.clear(); errline_ = 0; }
  void Clear() { DeleteChildren(); nodes_.clear(); attrs_.clear(); ClearError(); }

  XMLElement* NewElement(const char* name) {
    XMLElement* e = New<XMLElement>();
    e->value_ = name ? name : "";
    return e;
  }
  XMLComment* NewComment(const char* text) {
    XMLComment* c = New<XMLComment>();
    c->value_ = text ? text : "";
    return c;
  }
  XMLText* NewText(const char* text) {
    XMLText* t = New<XMLText>();
    t->value_ = text ? text : "";
    return t;
  }
  XMLAttribute* NewAttribute() {
    attrs_.emplace_back(new XMLAttribute());
    return attrs_.back().get();
  }

  XMLError Parse(const char* xml, size_t nbytes = (size_t)-1) {
    Clear();
    if (!xml) return SetError(XML_ERROR_EMPTY_DOCUMENT, 0, "null document");
    if (nbytes == (size_t)-1) nbytes = strlen(xml);
    src_.assign(xml, nbytes);
    // strip at embedded NUL
    size_t z = src_.find('\0');
    if (z != std::string::npos) src_.resize(z);
    pos_ = 0;
    curline_ = 1;
    if (src_.size() >= 3 && (unsigned char)src_[0] == 0xEF && (unsigned char)src_[1] == 0xBB &&
        (unsigned char)src_[2] == 0xBF) pos_ = 3;
    SkipSpace();
    if (pos_ >= src_.size()) return SetError(XML_ERROR_EMPTY_DOCUMENT, 0, "empty document");
    ParseChildren(this, nullptr);
    return err_;
  }
  XMLError LoadFile(const char* filename) {
    Clear();
    FILE* fp = fopen(filename, "rb");
    if (!fp) return SetError(XML_ERROR_FILE_NOT_FOUND, 0, filename);
    std::string data;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) data.append(buf, n);
    fclose(fp);
    return Parse(data.data(), data.size());
  }
  void Print(XMLPrinter* printer = nullptr) const {
    XMLPrinter stdoutPrinter(stdout);
    XMLPrinter* p = printer ? printer : &stdoutPrinter;
    for (const XMLNode* c = first_; c; c = c->NextSibling()) p->PrintNode(c, 0);
  }
  XMLError SaveFile(const char* filename, bool compact = false) {
    FILE* fp = fopen(filename, "w");
    if (!fp) return SetError(XML_ERROR_FILE_COULD_NOT_BE_OPENED, 0, filename);
    XMLPrinter printer(fp, compact);
    Print(&printer);
    fclose(fp);
    return XML_SUCCESS;
  }

 private:
  template <class T> T* New() {
    T* n = new T();
    n->doc_ = this;
    nodes_.emplace_back(n);
    return n;
  }
  XMLError SetError(XMLError e, int line, const std::string& what) {
    if (err_ == XML_SUCCESS) {
      err_ = e;
      errline_ = line;
      errstr_ = "Error=XML_ERROR_PARSING ErrorID=" + std::to_string((int)e) + " line=" +
                std::to_string(line) + ": " + what;
    }
    return err_;
  }
  bool Starts(const char* s) const { return src_.compare(pos_, strlen(s), s) == 0; }
  void Advance(size_t n) {
    for (size_t i = 0; i < n && pos_ < src_.size(); ++i) {
      if (src_[pos_] == '\n') ++curline_;
      ++pos_;
    }
  }
  void SkipSpace() {
    while (pos_ < src_.size() && isspace((unsigned char)src_[pos_])) Advance(1);
  }
  static bool IsNameChar(char c) {
    return isalnum((unsigned char)c) || c == '_' || c == ':' || c == '-' || c == '.' || ((unsigned char)c >= 128);
  }
  std::string ReadName() {
    size_t s = pos_;
    while (pos_ < src_.size() && IsNameChar(src_[pos_])) ++pos_;
    return src_.substr(s, pos_ - s);
  }
  static std::string Decode(const std::string& s) {
    std::string out;
    out.reserve(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
      if (s[i] == '\r') {  // normalise newlines
        if (i + 1 < s.size() && s[i + 1] == '\n') continue;
        out.push_back('\n');
        continue;
      }
      if (s[i] != '&') { out.push_back(s[i]); continue; }
      size_t semi = s.find(';', i);
      if (semi == std::string::npos || semi - i > 10) { out.push_back('&'); continue; }
      std::string ent = s.substr(i + 1, semi - i - 1);
      if (ent == "lt") out.push_back('<');
      else if (ent == "gt") out.push_back('>');
      else if (ent == "amp") out.push_back('&');
      else if (ent == "quot") out.push_back('"');
      else if (ent == "apos") out.push_back('\'');
      else if (!ent.empty() && ent[0] == '#') {
        unsigned long cp = (ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X'))
                               ? strtoul(ent.c_str() + 2, nullptr, 16)
                               : strtoul(ent.c_str() + 1, nullptr, 10);
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
          out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
          out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
          out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
          out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        }
      } else { out.append(s, i, semi - i + 1); }
      i = semi;
    }
    return out;
  }
  // parse children of `parent` until the closing tag named `closing` (or EOF at top level)
  void ParseChildren(XMLNode* parent, const char* closing) {
    while (!Error()) {
      // text run
      size_t s = pos_;
      int textline = curline_;
      size_t lt = src_.find('<', pos_);
      if (lt == std::string::npos) lt = src_.size();
      std::string text = src_.substr(s, lt - s);
      Advance(lt - s);
      bool allspace = true;
      for (char c : text) if (!isspace((unsigned char)c)) { allspace = false; break; }
      if (!allspace) {
        if (parent == this) { SetError(XML_ERROR_PARSING_TEXT, textline, "text outside root element"); return; }
        XMLText* t = NewText(Decode(text).c_str());
        t->line_ = textline;
        parent->InsertEndChild(t);
      }
      if (pos_ >= src_.size()) {
        if (closing) SetError(XML_ERROR_MISMATCHED_ELEMENT, curline_, std::string("missing </") + closing + ">");
        return;
      }
      int line = curline_;
      if (Starts("<!--")) {
        size_t e = src_.find("-->", pos_ + 4);
        if (e == std::string::npos) { SetError(XML_ERROR_PARSING_COMMENT, line, "unterminated comment"); return; }
        XMLComment* c = NewComment(src_.substr(pos_ + 4, e - pos_ - 4).c_str());
        c->line_ = line;
        parent->InsertEndChild(c);
        Advance(e + 3 - pos_);
      } else if (Starts("<![CDATA[")) {
        size_t e = src_.find("]]>", pos_ + 9);
        if (e == std::string::npos) { SetError(XML_ERROR_PARSING_CDATA, line, "unterminated CDATA"); return; }
        XMLText* t = NewText(src_.substr(pos_ + 9, e - pos_ - 9).c_str());
        t->line_ = line;
        parent->InsertEndChild(t);
        Advance(e + 3 - pos_);
      } else if (Starts("<?")) {
        size_t e = src_.find("?>", pos_ + 2);
        if (e == std::string::npos) { SetError(XML_ERROR_PARSING_DECLARATION, line, "unterminated declaration"); return; }
        Advance(e + 2 - pos_);
      } else if (Starts("<!")) {
        size_t e = src_.find('>', pos_ + 2);
        if (e == std::string::npos) { SetError(XML_ERROR_PARSING_UNKNOWN, line, "unterminated <! block"); return; }
        Advance(e + 1 - pos_);
      } else if (Starts("</")) {
        Advance(2);
        std::string name = ReadName();
        SkipSpace();
        if (pos_ >= src_.size() || src_[pos_] != '>') { SetError(XML_ERROR_PARSING_ELEMENT, line, "malformed closing tag"); return; }
        Advance(1);
        if (!closing || name != closing) {
          SetError(XML_ERROR_MISMATCHED_ELEMENT, line, "mismatched closing tag </" + name + ">");
        }
        return;
      } else {
        Advance(1);
        std::string name = ReadName();
        if (name.empty()) { SetError(XML_ERROR_PARSING_ELEMENT, line, "element name expected"); return; }
        XMLElement* e = NewElement(name.c_str());
        e->line_ = line;
        parent->InsertEndChild(e);
        XMLAttribute* lastattr = nullptr;
        bool selfclose = false;
        for (;;) {
          SkipSpace();
          if (pos_ >= src_.size()) { SetError(XML_ERROR_PARSING_ELEMENT, line, "unterminated element <" + name); return; }
          if (src_[pos_] == '>') { Advance(1); break; }
          if (Starts("/>")) { Advance(2); selfclose = true; break; }
          int aline = curline_;
          std::string an = ReadName();
          if (an.empty()) { SetError(XML_ERROR_PARSING_ATTRIBUTE, aline, "attribute name expected in <" + name + ">"); return; }
          SkipSpace();
          if (pos_ >= src_.size() || src_[pos_] != '=') { SetError(XML_ERROR_PARSING_ATTRIBUTE, aline, "'=' expected after attribute " + an); return; }
          Advance(1);
          SkipSpace();
          if (pos_ >= src_.size() || (src_[pos_] != '"' && src_[pos_] != '\'')) { SetError(XML_ERROR_PARSING_ATTRIBUTE, aline, "quoted value expected for attribute " + an); return; }
          char q = src_[pos_];
          size_t endq = src_.find(q, pos_ + 1);
          if (endq == std::string::npos) { SetError(XML_ERROR_PARSING_ATTRIBUTE, aline, "unterminated value for attribute " + an); return; }
          XMLAttribute* a = NewAttribute();
          a->name_ = an;
          a->value_ = Decode(src_.substr(pos_ + 1, endq - pos_ - 1));
          a->line_ = aline;
          Advance(endq + 1 - pos_);
          if (lastattr) lastattr->next_ = a; else e->attr_ = a;
          lastattr = a;
        }
        if (!selfclose) {
          ParseChildren(e, name.c_str());
          if (Error()) return;
        }
      }
    }
  }

  friend class XMLElement;
  std::vector<std::unique_ptr<XMLNode>> nodes_;
  std::vector<std::unique_ptr<XMLAttribute>> attrs_;
  XMLError err_ = XML_SUCCESS;
  std::string errstr_;
  int errline_ = 0;
  std::string src_;
  size_t pos_ = 0;
  int curline_ = 1;
};

inline XMLElement* XMLNode::FirstChildElement(const char* name) const {
  for (XMLNode* c = first_; c; c = c->next_) {
    XMLElement* e = c->ToElement();
    if (e && (!name || e->value_ == name)) return e;
  }
  return nullptr;
}
inline XMLElement* XMLNode::NextSiblingElement(const char* name) const {
  for (XMLNode* c = next_; c; c = c->next_) {
    XMLElement* e = c->ToElement();
    if (e && (!name || e->value_ == name)) return e;
  }
  return nullptr;
}
inline XMLNode* XMLNode::DeepClone(XMLDocument* target) const {
  XMLNode* clone = ShallowClone(target);
  if (!clone) return nullptr;
  for (const XMLNode* c = first_; c; c = c->next_) {
    XMLNode* cc = c->DeepClone(target);
    if (cc) clone->InsertEndChild(cc);
  }
  return clone;
}
inline XMLNode* XMLComment::ShallowClone(XMLDocument* target) const {
  XMLComment* c = (target ? target : doc_)->NewComment(Value());
  c->line_ = line_;
  return c;
}
inline XMLNode* XMLText::ShallowClone(XMLDocument* target) const {
  XMLText* t = (target ? target : doc_)->NewText(Value());
  t->line_ = line_;
  return t;
}
inline XMLNode* XMLElement::ShallowClone(XMLDocument* target) const {
  XMLDocument* d = target ? target : doc_;
  XMLElement* e = d->NewElement(Value());
  e->line_ = line_;
  XMLAttribute* last = nullptr;
  for (const XMLAttribute* a = attr_; a; a = a->next_) {
    XMLAttribute* na = d->NewAttribute();
    na->name_ = a->name_;
    na->value_ = a->value_;
    na->line_ = a->line_;
    if (last) last->next_ = na; else e->attr_ = na;
    last = na;
  }
  return e;
}
inline void XMLElement::SetAttribute(const char* name, const char* value) {
  XMLAttribute* last = nullptr;
  for (XMLAttribute* a = attr_; a; a = a->next_) {
    if (a->name_ == name) { a->value_ = value ? value : ""; return; }
    last = a;
  }
  XMLAttribute* a = doc_->NewAttribute();
  a->name_ = name;
  a->value_ = value ? value : "";
  if (last) last->next_ = a; else attr_ = a;
}
inline void XMLElement::SetText(const char* text) {
  for (XMLNode* c = first_; c; c = c->NextSibling()) {
    if (c->ToText()) { c->SetValue(text); return; }
  }
  InsertFirstChild(doc_->NewText(text));
}
inline void XMLPrinter::PrintNode(const XMLNode* n, int depth) {
  XMLNode* nn = const_cast<XMLNode*>(n);
  if (XMLElement* e = nn->ToElement()) {
    if (!compact_) PrintSpace(depth);
    Write("<");
    Write(e->Name());
    for (const XMLAttribute* a = e->FirstAttribute(); a; a = a->Next()) {
      Write(" ");
      Write(a->Name());
      Write("=\"");
      WriteEscaped(a->Value(), true);
      Write("\"");
    }
    if (e->NoChildren()) {
      Write("/>");
      if (!compact_) Write("\n");
      return;
    }
    bool onlytext = true;
    for (XMLNode* c = e->FirstChild(); c; c = c->NextSibling()) if (!c->ToText()) onlytext = false;
    Write(">");
    if (onlytext) {
      for (XMLNode* c = e->FirstChild(); c; c = c->NextSibling()) WriteEscaped(c->Value(), false);
    } else {
      if (!compact_) Write("\n");
      for (XMLNode* c = e->FirstChild(); c; c = c->NextSibling()) PrintNode(c, depth + 1);
      if (!compact_) PrintSpace(depth);
    }
    Write("</");
    Write(e->Name());
    Write(">");
    if (!compact_) Write("\n");
  } else if (nn->ToComment()) {
    if (!compact_) PrintSpace(depth);
    Write("<!--");
    Write(n->Value());
    Write("-->");
    if (!compact_) Write("\n");
  } else if (nn->ToText()) {
    if (!compact_) PrintSpace(depth);
    WriteEscaped(n->Value(), false);
    if (!compact_) Write("\n");
  }
}

}  // namespace tinyxml2

#endif  // ORACLE_STUB_TINYXML2_H_
