"""The static result page of the reference (util/html.py): `<web_dir>/index.html` with one header + one table row of linked
thumbnails per add_images() call, images under `<web_dir>/images/`.  The reference builds the document with `dominate`;
this writes the same elements (h3, table border=1 style="table-layout: fixed;", td > p > a > img + br + p) as plain text."""
import os
from html import escape


class HTML:
    def __init__(self, web_dir, title, reflesh=0):
        self.title = title
        self.web_dir = web_dir
        self.img_dir = os.path.join(self.web_dir, 'images')
        os.makedirs(self.img_dir, exist_ok=True)
        self._body = []
        self._refresh = int(reflesh)      # (the reference's spelling; it emits <meta http-equiv="reflesh" ...>)

    def get_image_dir(self):
        return self.img_dir

    def add_header(self, text):
        self._body.append('    <h3>%s</h3>' % escape(str(text)))

    def add_images(self, ims, txts, links, width=400):
        cells = []
        for im, txt, link in zip(ims, txts, links):
            cells.append('        <td style="word-wrap: break-word;" halign="center" valign="top">\n'
                         '          <p>\n'
                         '            <a href="%s"><img style="width:%dpx" src="%s"></a><br>\n'
                         '            <p>%s</p>\n'
                         '          </p>\n'
                         '        </td>' % (escape(os.path.join('images', link)), width, escape(os.path.join('images', im)), escape(str(txt))))
        self._body.append('    <table border="1" style="table-layout: fixed;">\n      <tr>\n%s\n      </tr>\n    </table>' % '\n'.join(cells))

    def render(self):
        head = '    <title>%s</title>' % escape(self.title)
        if self._refresh > 0:
            head += '\n    <meta http-equiv="reflesh" content="%d">' % self._refresh
        return '<!DOCTYPE html>\n<html>\n  <head>\n%s\n  </head>\n  <body>\n%s\n  </body>\n</html>\n' % (head, '\n'.join(self._body))

    def save(self):
        with open(os.path.join(self.web_dir, 'index.html'), 'wt') as f:
            f.write(self.render())
