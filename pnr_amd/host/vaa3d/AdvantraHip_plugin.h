// Vaa3D shell around libpnr_hip.so.
//
// NOT built in this repository's tests: it needs Qt4 and a Vaa3D source tree (v3d_interface.h), neither of which is
// available where the library is developed.  It exists so that
//     vaa3d -x Advantra -f advantra_func -i <img> -p <11 parameters>
// keeps working with the hot path on the GPU (the plugin is found by the name of the library, see AdvantraHip.pro).
// The five virtuals are what V3DPluginInterface2_1 asks of every plugin; only dofunc() does work here.
#pragma once
#include <QtGui>
#include <v3d_interface.h>

class AdvantraHipPlugin : public QObject, public V3DPluginInterface2_1 {
    Q_OBJECT
    Q_INTERFACES(V3DPluginInterface2_1);

  public:
    // command-line entry: "advantra_func" (input[0] = image files, input[1] = the 11 positional parameters) and "help"
    bool dofunc(const QString &function, const V3DPluginArgList &input, V3DPluginArgList &output, V3DPluginCallback2 &v3d, QWidget *parent);
    QStringList funclist() const;
    // menu entry: a notice that this build is driven from the command line
    void domenu(const QString &entry, V3DPluginCallback2 &v3d, QWidget *parent);
    QStringList menulist() const;
    float getPluginVersion() const { return 2.0f; }
};
